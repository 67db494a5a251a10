# block mode's shipped policy over batch shapes (+ its parity tests, twice)
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for i in 1 2; do timeout 600 python -m pytest tests/test_gpu_corpus.py -x -q -k "block or large_frames or blocks" 2>&1 | tail -1; done
run() { timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames $1 --frame-bytes $2 --gen-seconds 200 $4 2>/dev/null | pick "$1 x $(($2 >> 20)) MiB $3"; }
for cfg in "1 1073741824" "2 536870912" "4 268435456" "8 268435456" "1 67108864" "1 268435456" "3 134217728" "1 16777216" "16 134217728" "32 134217728" "64 134217728" "128 33554432"; do set -- $cfg; run $1 $2 shipped; done
run 1 1073741824 "shipped, offsets within 8 MiB" "--window-log 23"
run 64 134217728 "shipped, offsets within 8 MiB" "--window-log 23"
export MZD_LIB=$PWD/tmp_ab/libmzd_exp.so
MZD_EXP_BLK_GS=8 MZD_EXP_BLK_G=32 run 16 134217728 "gs=8 G=32"
MZD_EXP_BLK_GS=2 MZD_EXP_BLK_G=16 run 32 134217728 "gs=2 G=16"
