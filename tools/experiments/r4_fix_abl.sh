# what block mode's fix-up walk of MANY frames is made of (64 x 128 MiB): libraries with parts of k_blk_fixup removed (wrong bytes:
# timing only; built out of tree with -DMZD_ABL_FIX_{NOGATHER,NOSTORE,NOWAIT}, see profiles/r4_blk_fixup_ablations.txt for the patch)
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
run() { timeout 400 python bench.py --cpu-seconds 0 --no-ceiling --no-verify --steps 2 --warmup 1 --frames $1 --frame-bytes $2 --gen-seconds 200 2>/dev/null | pick "$1 x $(($2 >> 20)) MiB $3"; }
for m in "" NOGATHER NOSTORE NOWAIT ALL; do
  if [ -z "$m" ]; then unset MZD_LIB; else export MZD_LIB=$PWD/tmp_ab/libmzd_ablfix_$m.so; fi
  run 64 134217728 "${m:-shipped}"
  run 1 1073741824 "${m:-shipped}"
done
