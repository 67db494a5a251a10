cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
timeout 100 python bench.py --cpu-seconds 0 --no-ceiling --steps 10 2>/dev/null | pick "cfg4 default(q4)"
timeout 100 python bench.py --cpu-seconds 0 --no-ceiling --steps 10 --seq-variant 3 2>/dev/null | pick "cfg4 pipe"
timeout 100 python bench.py --cpu-seconds 0 --no-ceiling --steps 10 --seq-variant 1 2>/dev/null | pick "cfg4 k_seq"
