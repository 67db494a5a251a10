# k_exec_b alone at several residencies (extra dynamic LDS per frame): how few wavefronts per CU keep the stage fed?
cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for pad in 0 2400 5300 8400 12600 19400 32000; do
  timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 5 --exec-variant 2 --no-split --exec-chunk $pad 2>/dev/null | pick "k_exec_b pad=$pad"
done
