cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for w in 8 7 6 5; do
  L=$PWD/tmp_ab/libmzd_ew$w.so; [ $w = 8 ] && L=$PWD/sparkzstd_amd/libmzd.so
  MZD_LIB=$L timeout 100 python bench.py --cpu-seconds 0 --no-ceiling --steps 10 --no-split 2>/dev/null | pick "exec waves/SIMD=$w nosplit"
  MZD_LIB=$L timeout 100 python bench.py --cpu-seconds 0 --no-ceiling --steps 10 --seq-variant 2 2>/dev/null | pick "exec waves/SIMD=$w q4 split"
done
