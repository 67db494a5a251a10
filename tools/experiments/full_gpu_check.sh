# the GPU suite, the default bench line (with its `secondary` object), the fuzz soak and the chunk path's soak on the current sources
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out
timeout 2700 python3 -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r6_pytest_gpu.log
timeout 900 python3 bench.py > gpurun_out/r6_default_bench.json 2> gpurun_out/r6_default_bench.err; tail -c 600 gpurun_out/r6_default_bench.json
timeout 1500 python3 tools/fuzz_soak.py ${1:-60000} 5 2>&1 | tail -25 | tee gpurun_out/r6_fuzz_soak_final.txt
timeout 1200 python3 tools/chunk_soak.py ${2:-60000} 7 2>&1 | grep -E "DISAGREE|SOAK|0000 mutations" | tail -20 | tee gpurun_out/r6_chunk_soak.txt
