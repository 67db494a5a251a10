# round 3: the corpus through every variant (the work lists of a heterogeneous batch are ordered by size at upload), then the
# real-data bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_corpus.py tests/test_gpu_stages.py -x -q -k "not full_size and not bench_launches" 2>&1 | tail -6
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for v in 1 2; do
timeout 600 python bench.py --workload corpus --cpu-seconds 0 --no-ceiling --steps 10 --exec-variant $v 2>gpurun_out/r3_corpus.err | tee gpurun_out/r3_corpus_bench_v$v.json | pick "corpus exec_variant=$v"
done
tail -3 gpurun_out/r3_corpus.err
timeout 400 python bench.py --cpu-seconds 0 --no-ceiling --steps 5 2>/dev/null | pick "cfg4"
