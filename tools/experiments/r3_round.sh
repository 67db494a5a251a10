# round 3 evidence run: the whole GPU suite, the bench line of BASELINE config 4, the shard sizes of configs[4]
# (one GPU, the frames one rank of N gets), the real-data line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r3_pytest_gpu.log
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for n in 65536 32768 16384 8192; do
  timeout 400 python bench.py --cpu-seconds 0 --no-ceiling --steps 10 --frames $n 2>/dev/null | tee gpurun_out/r3_shard_$n.json | pick "frames=$n"
done
timeout 600 python bench.py --workload corpus --cpu-seconds 5 --steps 10 2>gpurun_out/r3_corpus.err | tee gpurun_out/r3_corpus_bench.json | pick "corpus"
tail -3 gpurun_out/r3_corpus.err
