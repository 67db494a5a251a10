# round 3: few large frames (the reference's own usage: one big frame per reader, cmd/sparkzstd/main.go:59,126) next to the batch
# of small ones: the same 8 GiB of output as 64 frames of 128 MiB and 8 GiB / 1 GiB as one frame; a single 64 MiB frame through
# the reader mirror; the streaming path on this round's kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 64 --frame-bytes 134217728 --gen-seconds 200 2>gpurun_out/r3_large64.err | tee gpurun_out/r3_large_64x128MiB.json | pick "64 x 128 MiB"
tail -2 gpurun_out/r3_large64.err
timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 1 --frame-bytes 1073741824 --gen-seconds 200 2>gpurun_out/r3_large1.err | tee gpurun_out/r3_large_1x1GiB.json | pick "1 x 1 GiB"
tail -2 gpurun_out/r3_large1.err
timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames 8192 --frame-bytes 1048576 2>/dev/null | tee gpurun_out/r3_large_8192x1MiB.json | pick "8192 x 1 MiB"
timeout 600 python tools/stream_bench.py 8192 12 1,2,3 2>/dev/null | tail -3 | tee gpurun_out/r3_stream.json
timeout 300 python tools/reader_bench.py 256 2>/dev/null | tail -2 | tee gpurun_out/r3_reader.txt
