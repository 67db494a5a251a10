cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "decodecorpus_bit_exact_on_gpu or oracle_trace or fuzzed_frames or escape_codes or randomized or synthetic_configs or multi_block or window_by_window or split_batch" 2>&1 | tail -4
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for r in 1 2; do
MZD_LIB=$PWD/tmp_ab/libmzd_before_early.so timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 10 2>/dev/null | pick "cfg4 pipe OLD"
timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 10 2>/dev/null | pick "cfg4 pipe EARLY-READS"
done
timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 10 --no-split 2>/dev/null | pick "cfg4 pipe EARLY-READS nosplit"
