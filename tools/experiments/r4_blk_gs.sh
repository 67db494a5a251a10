# block mode, blocks per job (MZD_EXP_BLK_GS) x fix-up workgroups per frame (MZD_EXP_BLK_G): fewer, longer jobs = slower passes, shorter walk
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
export MZD_LIB=$PWD/tmp_ab/libmzd_exp.so
for gs in 2 4 8 16; do
  MZD_EXP_BLK_GS=$gs timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames 1 --frame-bytes 1073741824 --gen-seconds 200 2>/dev/null | pick "1 x 1 GiB gs=$gs"
done
for gs in 1 2 4; do
  MZD_EXP_BLK_GS=$gs timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames 8 --frame-bytes 268435456 --gen-seconds 200 2>/dev/null | pick "8 x 256 MiB gs=$gs"
done
for gs in 1 2 4; do
  MZD_EXP_BLK_GS=$gs timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 64 --frame-bytes 134217728 --gen-seconds 200 2>/dev/null | pick "64 x 128 MiB gs=$gs"
done
