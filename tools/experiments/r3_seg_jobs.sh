# round 3: block mode -- jobs of several consecutive blocks, k_blk_scan as a scan: parity, then the large-frame lines
cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
timeout 1500 python -m pytest tests/test_gpu_corpus.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 1 --frame-bytes 1073741824 --gen-seconds 200 2>/dev/null | pick "1 x 1 GiB"
timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 64 --frame-bytes 134217728 --gen-seconds 200 2>/dev/null | pick "64 x 128 MiB"
timeout 900 python tools/fuzz_soak.py 6000 3 2>&1 | tail -2
