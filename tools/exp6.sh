cd $GRAFT_REPO_ROOT
B="python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-ceiling"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
export MZD_LIB=$PWD/tmp_ab/libmzd_prio.so
$B 2>/dev/null | pick "prio build, normal pass"
for x in 40,5,256 44,4,256 40,1,256 40,2,256 40,3,256; do MZD_EXP_OVERLAP=$x $B 2>/dev/null | pick "prio overlap nch,R,thr=$x"; done
