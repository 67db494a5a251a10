# round 3: heterogeneous batches with the Huffman kernel beside the sequence stage (second stream) instead of before it
cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for i in 1 2; do timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --workload corpus --steps 5 --warmup 2 2>/dev/null | pick corpus; done
[ "$1" = "notests" ] || timeout 900 python -m pytest tests/test_gpu_corpus.py -m gpu -x -q 2>&1 | tail -3
