# block mode after a change to the fix-up walk: its parity tests (three times: the protocol depends on timing), then the two
# large-frame lines
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for i in 1; do timeout 600 python -m pytest tests/test_gpu_corpus.py -x -q -k "block or large_frames or blocks" 2>&1 | tail -2; done
timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames 1 --frame-bytes 1073741824 --gen-seconds 200 2>/dev/null | pick "1 x 1 GiB"
timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 64 --frame-bytes 134217728 --gen-seconds 200 2>/dev/null | pick "64 x 128 MiB"
# the same frames with matches that stay within 8 MiB (zstd's windowLog 23): three passes instead of four
timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames 1 --frame-bytes 1073741824 --gen-seconds 200 --window-log 23 2>/dev/null | pick "1 x 1 GiB, offsets <= 8 MiB"
timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 64 --frame-bytes 134217728 --gen-seconds 200 --window-log 23 2>/dev/null | pick "64 x 128 MiB, offsets <= 8 MiB"
