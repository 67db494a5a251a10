# the GPU suite + the headline + configs 2 / 3 on the current tree
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'], d['roofline'].get('frac'), d['roofline'].get('copy_ceiling'))" "$1"; }
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
timeout 600 python bench.py --cpu-seconds 0 --steps 10 2>/dev/null | pick "config 4"
timeout 300 python bench.py --cpu-seconds 0 --config 2 --steps 20 2>/dev/null | pick "config 2"
timeout 300 python bench.py --cpu-seconds 0 --config 3 --steps 20 2>/dev/null | pick "config 3"
timeout 300 python tools/reader_bench.py 1024 2>&1 | tail -5
