cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "decodecorpus_bit_exact_on_gpu or oracle_trace or fuzzed_frames or escape_codes or randomized or synthetic_configs or multi_block or window_by_window" 2>&1 | tail -4
MZD_LIB=$PWD/tmp_ab/libmzd_q4stats.so timeout 300 python tools/q4_stats.py 13824
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --seq-variant 2 --steps 5 --no-split 2>/dev/null | pick "cfg4 q4(asm) nosplit"
