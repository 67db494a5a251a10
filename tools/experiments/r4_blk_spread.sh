# block mode, few frames: the fix-up workgroups of a frame over ALL XCDs (MZD_EXP_BLK_SPREAD=1) instead of on one, G per frame, gs blocks per job
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
export MZD_LIB=$PWD/tmp_ab/libmzd_exp.so
run() { # frames frame_bytes spread G gs
  MZD_EXP_BLK_SPREAD=$3 MZD_EXP_BLK_G=$4 MZD_EXP_BLK_GS=$5 timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames $1 --frame-bytes $2 --gen-seconds 200 2>/dev/null | pick "$1 x $(($2 >> 20)) MiB spread=$3 G=$4 gs=$5"
}
for cfg in "64 4" "96 4" "128 4" "160 4" "128 3" "96 3" "128 6" "192 6" "128 5"; do set -- $cfg; run 1 1073741824 1 $1 $2; done
for cfg in "32 4" "64 4" "128 4" "64 2"; do set -- $cfg; run 2 536870912 1 $1 $2; done
for cfg in "16 4" "32 4" "64 4" "32 2"; do set -- $cfg; run 4 268435456 1 $1 $2; done
run 4 268435456 0 64 2
run 2 536870912 0 64 2
for cfg in "2 1" "4 1" "8 1" "4 2"; do set -- $cfg; run 64 134217728 1 $1 $2; done
