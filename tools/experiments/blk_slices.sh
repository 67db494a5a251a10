# block mode in slices: the parity tests of block mode on the shipped library (incl. test_block_mode_in_slices), then K = 1 / 2 / 4 / 8 on
# few large frames with the experiments library (tmp_ab/libmzd_exp.so, built with -DMZD_EXPERIMENTS), one generated batch per shape
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "block_mode or blocks_ or multi_block or large" 2>&1 | tail -4
export MZD_LIB=$PWD/tmp_ab/libmzd_exp.so
timeout 1500 python3 tools/experiments/blk_slices.py "$@"
