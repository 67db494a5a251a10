cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 5 --warmup 2 2>/dev/null | pick cfg4
timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --frames 8192 2>/dev/null | pick shard8192
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
