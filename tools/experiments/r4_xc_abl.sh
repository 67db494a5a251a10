# k_exec_c ablations (timing only; wrong bytes): what the kernel's time is made of
pick() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', 'k_exec', d['roofline']['kernel_ms'].get('k_exec'))"; }
timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 4 --exec-variant 5 --no-split --no-verify 2>/dev/null | pick "shipped"
for l in NOWAIT NOSTAGE NOFLUSH NOPASS SETUPONLY; do
  MZD_LIB=$PWD/tmp_ab/libmzd_abl_$l.so timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 4 --exec-variant 5 --no-split --no-verify 2>/dev/null | pick "$l"
done
