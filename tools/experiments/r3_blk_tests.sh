cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_corpus.py -m gpu -x -q -k "block_mode or large_frames" 2>&1 | tail -15
