cd $GRAFT_REPO_ROOT
MZD_DEBUG_SEQ_ONLY=1 python tools/seq_diff.py z000026 z000088 z000070 z000000
timeout 900 python -m pytest tests -m gpu -x -q -k "decodecorpus_bit_exact_on_gpu or oracle_trace or fuzzed_frames or escape_codes or randomized or synthetic_configs or multi_block or window_by_window" 2>&1 | tail -4
