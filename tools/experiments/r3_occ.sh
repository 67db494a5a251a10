# k_exec_b alone at several residencies (extra dynamic LDS per frame): frames in flight vs the cache footprint of their slabs;
# and the same with the stretch's loads travelling behind its passes (-DMZD_ABL_XB_LATE: wrong bytes, timing only)
cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for pad in 0 2000 5400 8800 15700 36000; do
  timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 5 --exec-variant 2 --no-split --exec-chunk $pad 2>/dev/null | pick "k_exec_b pad=$pad"
  [ -f tmp_ab/libmzd_xblate.so ] && MZD_LIB=$PWD/tmp_ab/libmzd_xblate.so timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 5 --exec-variant 2 --no-split --exec-chunk $pad --no-verify 2>/dev/null | pick "k_exec_b LATE pad=$pad"
done
