cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for v in 2 4; do
  MZD_LIB=$PWD/tmp_ab/libmzd_sl$v.so timeout 100 python bench.py --cpu-seconds 0 --no-ceiling --steps 10 --no-split 2>/dev/null | pick "idle sleep $v nosplit"
  MZD_LIB=$PWD/tmp_ab/libmzd_sl$v.so timeout 100 python bench.py --cpu-seconds 0 --no-ceiling --steps 10 2>/dev/null | pick "idle sleep $v split"
done
for t in 64 128 256; do for c in 4096 8192 16384; do
  timeout 100 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --no-split --exec-threads $t --exec-chunk $c 2>/dev/null | pick "threads $t chunk $c nosplit"
done; done
timeout 100 python bench.py --cpu-seconds 0 --no-ceiling --steps 10 2>/dev/null | pick "base split"
