cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for v in base NOWT NOFAR NOLIT; do
  L=$PWD/tmp_ab/libmzd_x$v.so; [ $v = base ] && L=$PWD/sparkzstd_amd/libmzd.so
  MZD_LIB=$L timeout 100 python bench.py --cpu-seconds 0 --no-ceiling --steps 10 --seq-variant 2 --no-split 2>/dev/null | pick "exec ablation $v nosplit"
done
