# block mode with all passes in one launch: blocks per job (gs) x fix-up workgroups per frame (G), by batch shape
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
export MZD_LIB=$PWD/tmp_ab/libmzd_exp.so
run() { MZD_EXP_BLK_GS=$3 MZD_EXP_BLK_G=$4 timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames $1 --frame-bytes $2 --gen-seconds 200 2>/dev/null | pick "$1 x $(($2 >> 20)) MiB gs=$3 G=$4"; }
for c in "16 256" "12 256" "8 128" "8 192" "10 256"; do set -- $c; run 1 1073741824 $1 $2; done
for c in "8 128" "8 64" "4 64"; do set -- $c; run 2 536870912 $1 $2; done
for c in "4 32" "4 64" "8 64" "8 32"; do set -- $c; run 4 268435456 $1 $2; done
for c in "8 32" "8 64" "4 32"; do set -- $c; run 8 268435456 $1 $2; done
for c in "4 128" "8 256" "4 64"; do set -- $c; run 1 268435456 $1 $2; done
for c in "2 64" "4 128" "2 32"; do set -- $c; run 1 67108864 $1 $2; done
for c in "2 16" "2 32" "4 32"; do set -- $c; run 16 134217728 $1 $2; done
