cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for v in 2 1; do timeout 300 python bench.py --workload corpus --cpu-seconds 0 --no-ceiling --steps 8 --exec-variant $v 2>/dev/null | pick "corpus exec_variant=$v"; done
timeout 900 python -m pytest tests/test_gpu_corpus.py tests/test_gpu_stages.py -x -q -k "not full_size and not bench_launches" 2>&1 | tail -3
