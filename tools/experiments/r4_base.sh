# round 4 baseline: config-4 pass with either execution kernel, split and alone; k_exec_b tile statistics
pick() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', d['ms_per_step'], d['roofline']['kernel_ms'], 'bit_exact', d.get('bit_exact'))"; }
for v in 1 2; do
  timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --exec-variant $v 2>/dev/null | pick "split exec_variant=$v"
  timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --exec-variant $v --no-split 2>/dev/null | pick "no-split exec_variant=$v"
done
MZD_LIB=$PWD/tmp_ab/libmzd_xbstats.so timeout 300 python tools/xb_stats.py 16384 2>&1 | tail -16
