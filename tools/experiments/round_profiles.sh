# The evidence of a round, parameterised by its tag (profiles/<tag>_*): the GPU suite, then rocprofv3 kernel stats + FETCH / WRITE
# traffic + SQ counters + the bench line (quoting them) for BASELINE config 4, the real-data workload, config 3 and the 8192-frame
# shard of configs[4]; the config-2 bench line; few large frames (block mode) with kernel stats; the readers; the streaming path; the
# shard table; one frame in chunks.  usage: bash tools/experiments/round_profiles.sh r5 [quick]     (quick: config 4 only)
TAG=${1:-r6}
QUICK=${2:-}
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
bash tools/profile_counters.sh $TAG 4 2>&1 | tail -2
bash tools/profile_round.sh $TAG 4 --issue-from gpurun_out/${TAG}_issue_cfg4.json 2>&1 | tail -1 | cut -c1-400
if [ -n "$QUICK" ]; then exit 0; fi
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/${TAG}_pytest_gpu.log
bash tools/profile_counters.sh ${TAG}corpus 4 --workload corpus 2>&1 | tail -2
bash tools/profile_round.sh ${TAG}corpus 4 --workload corpus --issue-from gpurun_out/${TAG}corpus_issue_cfg4.json 2>&1 | tail -1 | cut -c1-400
bash tools/profile_round.sh ${TAG}s8192 4 --frames 8192 2>&1 | tail -1 | cut -c1-300
bash tools/profile_round.sh $TAG 3 2>&1 | tail -1 | cut -c1-300
bash tools/profile_round.sh $TAG 2 2>&1 | tail -1 | cut -c1-300
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/${TAG}_blk_prof
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_blk_prof --output-format csv -- python3 bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 64 --frame-bytes 134217728 --gen-seconds 200 2>gpurun_out/${TAG}_blk64.err | tee gpurun_out/${TAG}_large_64x128MiB_bench.json | pick "64 x 128 MiB"
find gpurun_out/${TAG}_blk_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_large_64x128MiB_kernel_stats.csv
rm -rf gpurun_out/${TAG}_blk_prof
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_blk_prof --output-format csv -- python3 bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 1 --frame-bytes 1073741824 --gen-seconds 200 2>gpurun_out/${TAG}_blk1.err | tee gpurun_out/${TAG}_large_1x1GiB_bench.json | pick "1 x 1 GiB"
find gpurun_out/${TAG}_blk_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_large_1x1GiB_kernel_stats.csv
rm -rf gpurun_out/${TAG}_blk_prof
for cfg in "4 268435456" "16 134217728"; do set -- $cfg; timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames $1 --frame-bytes $2 --gen-seconds 200 2>/dev/null | tee gpurun_out/${TAG}_large_${1}x$(($2 >> 20))MiB_bench.json | pick "$1 x $(($2 >> 20)) MiB"; done
timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 64 --frame-bytes 134217728 --gen-seconds 200 --window-log 23 2>/dev/null | tee gpurun_out/${TAG}_large_64x128MiB_wlog23_bench.json | pick "64 x 128 MiB, offsets within 8 MiB"
timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames 8192 --frame-bytes 1048576 2>/dev/null | tee gpurun_out/${TAG}_large_8192x1MiB_bench.json | pick "8192 x 1 MiB"
timeout 600 python tools/stream_bench.py 8192 12 1,2,3 2>/dev/null | tail -3 | tee gpurun_out/${TAG}_stream.json
timeout 600 python tools/reader_bench.py 1024 67108864 2>/dev/null | tail -6 | tee gpurun_out/${TAG}_reader.txt
timeout 600 python tools/reader_bench_cpp.py 2048 32768 2>/dev/null | tee gpurun_out/${TAG}_reader_cpp.jsonl | cut -c1-200
bash tools/experiments/large_reader.sh 2>&1 | tee gpurun_out/${TAG}_large_frame_reader.txt
# one frame in chunks (mzd_fstream_*, ABI 9): 1 GiB with a window of 8 MiB, 2.5 GiB (beyond block mode's whole frames) with one of 128 MiB
(timeout 600 python tools/fstream_bench.py 1024 23 4,16,64,256; timeout 900 python tools/fstream_bench.py 2560 27 64,256) 2>/dev/null | tee gpurun_out/${TAG}_fstream.jsonl | cut -c1-330
for n in 65536 32768 16384 8192; do timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --frames $n 2>/dev/null | tee gpurun_out/${TAG}_shard_${n}_1gpu.json | pick "shard $n"; done
ls gpurun_out | grep "^${TAG}" | head -80
