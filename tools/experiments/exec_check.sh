# k_exec_c variants: the execution-related GPU tests with EACH library under tmp_ab named in CHECK_LIBS, then same-box A/B of all named libraries
cd ${GRAFT_REPO_ROOT:-$PWD}
for l in $CHECK_LIBS; do echo "== tests with $l"; MZD_LIB=$PWD/tmp_ab/$l timeout 1500 python3 -m pytest tests -m gpu -x -q -k "decodecorpus_bit_exact_on_gpu or stage_boundaries or oracle_trace or fuzz or corrupt or truncat or raw_rle or config4 or synthetic or block_mode or multi_block or periodic or overlap" 2>&1 | tail -3; done
bash tools/experiments/ab.sh "$@"
