# round 3: the shards of configs[4] -- the Huffman kernel beside the sequence stage when that stage is a single round
cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for n in 4096 8192 16384 32768 65536; do timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --frames $n 2>/dev/null | pick "frames $n"; done
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_corpus.py -m gpu -x -q -k "bench or split or config4 or synthetic" 2>&1 | tail -3
