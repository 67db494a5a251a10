# usage: r4_xc_quick.sh [lib ...]   (libraries under tmp_ab/; none: the shipped one)
pick() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', 'pass', d['ms_per_step'], d['roofline']['kernel_ms'], 'bit_exact', d.get('bit_exact'))"; }
one() {
  timeout 150 python tools/xc_debug.py 64 2>&1 | tail -2
  timeout 200 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --exec-variant 5 --no-split 2>/dev/null | pick "$1 no-split exec_variant=5"
  timeout 200 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --exec-variant 5 2>/dev/null | pick "$1 split exec_variant=5"
}
if [ $# -eq 0 ]; then one shipped; fi
for l in "$@"; do export MZD_LIB=$PWD/tmp_ab/$l; one $l; done
