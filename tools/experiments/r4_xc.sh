# k_exec_c (exec_variant 5): parity on the corpus and the synthetic suites, then config 4 against k_exec / k_exec_b
pick() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', 'pass', d['ms_per_step'], d['roofline']['kernel_ms'], 'bit_exact', d.get('bit_exact'))"; }
timeout 600 python tools/xc_debug.py 64 2>&1 | tail -12
timeout 1500 python -m pytest tests/test_gpu_corpus.py -x -q -k "k_exec_c or 5" 2>&1 | tail -8
for v in 5; do
  timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --exec-variant $v --no-split 2>/dev/null | pick "no-split exec_variant=$v"
done
timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --exec-variant 5 2>/dev/null | pick "split exec_variant=5"
MZD_LIB=$PWD/tmp_ab/libmzd_xcstats.so timeout 300 python tools/xc_stats.py 16384 2>&1 | tail -16
