# round 3 evidence: the GPU suite, then rocprofv3 kernel stats + FETCH / WRITE traffic + SQ counters + the bench line (quoting
# them) for BASELINE config 4, the real-data workload, config 3 and the 8192-frame shard of configs[4]; config 2 bench line; few
# large frames (block mode) with kernel stats; the reader mirror on one 64 MiB frame; the streaming path
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r3_pytest_gpu.log
bash tools/profile_counters.sh r3 4 2>&1 | tail -2
bash tools/profile_round.sh r3 4 --issue-from gpurun_out/r3_issue_cfg4.json 2>&1 | tail -1 | cut -c1-400
bash tools/profile_counters.sh r3corpus 4 --workload corpus 2>&1 | tail -2
bash tools/profile_round.sh r3corpus 4 --workload corpus --issue-from gpurun_out/r3corpus_issue_cfg4.json 2>&1 | tail -1 | cut -c1-400
bash tools/profile_round.sh r3s8192 4 --frames 8192 2>&1 | tail -1 | cut -c1-300
bash tools/profile_round.sh r3 3 2>&1 | tail -1 | cut -c1-300
timeout 300 python bench.py --config 2 2>/dev/null | tee gpurun_out/r3_cfg2_bench.json | cut -c1-200
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r3_blk_prof
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r3_blk_prof --output-format csv -- python3 bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 64 --frame-bytes 134217728 --gen-seconds 200 2>gpurun_out/r3_blk64.err | tee gpurun_out/r3_large_64x128MiB_bench.json | pick "64 x 128 MiB"
find gpurun_out/r3_blk_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3_large_64x128MiB_kernel_stats.csv
rm -rf gpurun_out/r3_blk_prof
timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 1 --frame-bytes 1073741824 --gen-seconds 200 2>/dev/null | tee gpurun_out/r3_large_1x1GiB_bench.json | pick "1 x 1 GiB"
timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames 8192 --frame-bytes 1048576 2>/dev/null | tee gpurun_out/r3_large_8192x1MiB_bench.json | pick "8192 x 1 MiB"
timeout 600 python tools/stream_bench.py 8192 12 1,2,3 2>/dev/null | tail -3 | tee gpurun_out/r3_stream.json
timeout 600 python tools/reader_bench.py 256 67108864 2>/dev/null | tail -4 | tee gpurun_out/r3_reader.txt
for n in 65536 32768 16384 8192; do timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --frames $n 2>/dev/null | tee gpurun_out/r3_shard_${n}_1gpu.json | pick "shard $n"; done
ls gpurun_out | grep "^r3" | head -80
