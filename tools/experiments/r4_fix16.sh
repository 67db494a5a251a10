# the fix-up walk with one 16-byte gather for a chunk's dominant distance (shipped) against four-byte gathers only (tmp_ab/libmzd_prev.so), same box
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for i in 1 2; do timeout 600 python -m pytest tests/test_gpu_corpus.py -x -q -k "block or large_frames or blocks" 2>&1 | tail -1; done
run() { timeout 400 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames $1 --frame-bytes $2 --gen-seconds 200 2>/dev/null | pick "$1 x $(($2 >> 20)) MiB $3"; }
for cfg in "64 134217728" "16 134217728" "1 1073741824" "4 268435456" "1 67108864"; do
  set -- $cfg
  unset MZD_LIB; run $1 $2 "16-byte gather"
  export MZD_LIB=$PWD/tmp_ab/libmzd_prev.so; run $1 $2 "four-byte gathers"
done
