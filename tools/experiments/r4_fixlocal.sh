# (-DMZD_FIX_LOCAL was never committed: two lines of mzd_exec_blk.hip, see profiles/r4_blk_fixup_local_l2.txt)
# block mode's fix-up walk with the hand-off kept inside the frame's XCD (plain stores, the counter's atomic without scope bits: the
# lines stay in that L2; gathers bypass L1 only) against write-through stores and agent-scope atomics
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for lib in $PWD/tmp_ab/libmzd_fixlocal.so; do
  export MZD_LIB=$lib
  [ -z "$lib" ] && unset MZD_LIB
  timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames 1 --frame-bytes 1073741824 --gen-seconds 200 2>/dev/null | pick "1 x 1 GiB lib=$lib"
  timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 64 --frame-bytes 134217728 --gen-seconds 200 2>/dev/null | pick "64 x 128 MiB lib=$lib"
  timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames 8 --frame-bytes 268435456 --gen-seconds 200 2>/dev/null | pick "8 x 256 MiB lib=$lib"
done
