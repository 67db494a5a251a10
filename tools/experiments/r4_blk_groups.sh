# block mode, 2-16 frames: the batch in two groups of frames, the passes of the second beside the fix-up walk of the first (MZD_EXP_BLK_GROUPS=1)
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
export MZD_LIB=$PWD/tmp_ab/libmzd_exp.so
MZD_EXP_BLK_GROUPS=1 timeout 900 python -m pytest tests/test_gpu_corpus.py -x -q -k "block or large or policy" 2>&1 | tail -1
run() { timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames $1 --frame-bytes $2 --gen-seconds 200 2>/dev/null | pick "$1 x $(($2 >> 20)) MiB $3"; }
for cfg in "2 536870912" "4 268435456" "8 268435456" "16 134217728" "3 134217728" "2 67108864"; do
  set -- $cfg
  run $1 $2 "one group"
  MZD_EXP_BLK_GROUPS=1 run $1 $2 "two groups"
done
