# round 3: fix-up gathers grouped by distance per dword
cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['kernel_ms']['k_exec'], d['bit_exact'])" "$1"; }
timeout 900 python -m pytest tests/test_gpu_corpus.py -m gpu -x -q -k "blocks or segments or block_mode or large_frames or 0-0-3 or 0-0-4 or 0-3 or 0-4" 2>&1 | tail -1
timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 64 --frame-bytes 134217728 --gen-seconds 200 2>/dev/null | pick "64 x 128 MiB"
timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 1 --frame-bytes 1073741824 --gen-seconds 200 2>/dev/null | pick "1 x 1 GiB"
