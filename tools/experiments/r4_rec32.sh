pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
timeout 600 python tools/xc_debug.py 64 2>&1 | tail -2
timeout 2000 python -m pytest tests/test_gpu_corpus.py tests/test_gpu_stages.py -x -q 2>&1 | tail -4
timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 8 2>/dev/null | pick "config 4"
timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 8 --no-split 2>/dev/null | pick "config 4 no-split"
timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 8 --frames 8192 2>/dev/null | pick "shard 8192"
timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 4 --workload corpus 2>/dev/null | pick "corpus"
timeout 300 python bench.py --cpu-seconds 0 --config 2 --steps 20 2>/dev/null | pick "config 2"
MZD_LIB=$PWD/tmp_ab/libmzd_xcstats.so timeout 300 python tools/xc_stats.py corpus 40 2>&1 | tail -16
