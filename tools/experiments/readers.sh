#!/bin/bash
# GPU box: the readers host to host (round 6) -- the reader tests, the C++ BatchFrameReader over mzd_stream_* (tools/reader_bench_cpp.py),
# the Python readers (tools/reader_bench.py), the streaming path itself (tools/stream_bench.py).  usage: tools/experiments/readers.sh
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -m pytest tests/test_gpu_corpus.py -m gpu -x -q -k "reader or verify_cli or stream" 2>&1 | tail -3
echo "== C++ BatchFrameReader over mzd_stream_* (tools/verify/sparkzstd_verify --bench)"
python3 tools/reader_bench_cpp.py 2048 32768 2>&1 | grep -v amdgpu.ids
echo "== Python readers (tools/reader_bench.py 2048)"
python3 tools/reader_bench.py 2048 2>&1 | grep -v amdgpu.ids
echo "== mzd_stream_* itself (tools/stream_bench.py 8192 8 2)"
python3 tools/stream_bench.py 8192 8 2 2>&1 | grep -v amdgpu.ids
