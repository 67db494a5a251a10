cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -k "huf_seg or oracle_trace or config3 or config2" 2>&1 | tail -15
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'], d['roofline'].get('frac_of_copy_ceiling'))" "$1"; }
python bench.py --config 3 --cpu-seconds 0 2>/dev/null | pick "cfg3 auto(seg)"
python bench.py --config 3 --cpu-seconds 0 --huf-variant 1 2>/dev/null | pick "cfg3 lane"
python bench.py --config 2 --cpu-seconds 0 2>/dev/null | pick "cfg2"
python bench.py --cpu-seconds 0 --no-ceiling --huf-variant 2 2>/dev/null | pick "cfg4 seg"
python bench.py --cpu-seconds 0 --no-ceiling 2>/dev/null | pick "cfg4 lane"
