cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_corpus.py -m gpu -x -q -k "huf or config3 or literals_and_sequences or decodecorpus or fuzz or corrupt or randomized" 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --config 3 --cpu-seconds 0 --no-ceiling 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['bit_exact'])"; done
timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --workload corpus --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('corpus', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['bit_exact'])"
