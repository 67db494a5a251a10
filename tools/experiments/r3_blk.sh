# round 3: blocks of a frame executed side by side (mzd_exec_blk.hip) -- parity first, then the few-large-frames lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
[ "$1" = "perf" ] || timeout 900 python -m pytest tests/test_gpu_corpus.py -m gpu -x -q -k "blocks or 0-0-3 or 3-0-3 or 0-3" 2>&1 | tail -15 | tee gpurun_out/r3_blk_pytest.log
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
if [ "$1" != "tests" ]; then
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r3_blk_prof --output-format csv -- python3 bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 64 --frame-bytes 134217728 --gen-seconds 200 2>gpurun_out/r3_blk64.err | tee gpurun_out/r3_large_64x128MiB.json | pick "64 x 128 MiB"
tail -2 gpurun_out/r3_blk64.err
find gpurun_out/r3_blk_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3_blk64_kernel_stats.csv
head -12 gpurun_out/r3_blk64_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/r3_blk_prof
[ "$2" = "only64" ] || timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 1 --frame-bytes 1073741824 --gen-seconds 200 2>gpurun_out/r3_blk1.err | tee gpurun_out/r3_large_1x1GiB.json | pick "1 x 1 GiB"
tail -2 gpurun_out/r3_blk1.err
[ "$2" = "cfg4" ] && timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 2>/dev/null | pick "cfg4"
[ "$2" = "cfg4" ] && timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --exec-variant 2 2>/dev/null | pick "cfg4 k_exec_b"
fi
