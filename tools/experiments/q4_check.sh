cd $GRAFT_REPO_ROOT
MZD_DEBUG_SEQ_ONLY=1 timeout 60 python tools/seq_diff.py z000026 z000088 z000070 z000000 z000003 || exit 1
timeout 200 python -m pytest tests -m gpu -x -q -k "decodecorpus_bit_exact_on_gpu or oracle_trace or fuzzed_frames or escape_codes or randomized or synthetic_configs or multi_block or window_by_window" 2>&1 | tail -4
MZD_LIB=$PWD/tmp_ab/libmzd_q4stats.so timeout 100 python tools/q4_stats.py 13824
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
timeout 100 python bench.py --cpu-seconds 0 --no-ceiling --seq-variant 2 --steps 10 2>/dev/null | pick "cfg4 q4"
timeout 100 python bench.py --cpu-seconds 0 --no-ceiling --seq-variant 2 --steps 10 --no-split 2>/dev/null | pick "cfg4 q4 nosplit"
timeout 100 python bench.py --cpu-seconds 0 --no-ceiling --steps 10 --seq-variant 3 2>/dev/null | pick "cfg4 pipe"
MZD_LIB=$PWD/tmp_ab/libmzd_q4prof.so timeout 100 python tools/q4_stats.py 13824 2>&1 | grep -E "^B|^C" | head -4
