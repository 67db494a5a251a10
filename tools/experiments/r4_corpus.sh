# the real-data workload: k_huf's large-table class from global memory (MZD_EXP_HUF_GT = first class that does: 3 = none, 2 default, 1)
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for g in 3 2 1 0; do
  MZD_LIB=$PWD/tmp_ab/libmzd_exp.so MZD_EXP_HUF_GT=$g timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 4 --workload corpus 2>/dev/null | pick "corpus, classes >= $g from global tables"
done
timeout 900 python -m pytest tests/test_gpu_corpus.py -x -q -k "decodecorpus or device_planner or fuzz" 2>&1 | tail -3
