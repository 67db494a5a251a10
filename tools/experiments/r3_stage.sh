# round 3: k_seq_q4 table staging per chain (a wavefront per chain, the cells the tables really have)
cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_corpus.py -m gpu -x -q -k "literals_and_sequences or decodecorpus or escape or small_frames or randomized or device_planner_corpus" 2>&1 | tail -3
for i in 1 2; do timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --workload corpus --steps 5 --warmup 2 2>/dev/null | pick corpus; done
for i in 1 2; do timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 5 --warmup 2 2>/dev/null | pick cfg4; done
timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --frames 8192 2>/dev/null | pick shard8192
timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --frames 131072 --frame-bytes 4096 2>/dev/null | pick 4KiBframes
