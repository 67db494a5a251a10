# same-box A/B of libraries under tmp_ab on config 4 (k_seq / k_exec alone with --no-split, and the split pass), alternating twice
cd ${GRAFT_REPO_ROOT:-$PWD}
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for rep in 1 2; do
  for l in "$@"; do
    export MZD_LIB=$PWD/tmp_ab/$l
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 8 --no-split $BENCH_EXTRA 2>/dev/null | pick "$l no-split"
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 8 $BENCH_EXTRA 2>/dev/null | pick "$l split"
  done
done
