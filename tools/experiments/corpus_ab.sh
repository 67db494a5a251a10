# real data (the reference's corpus, replicated): same-box A/B of libraries under tmp_ab at 1 and 4 GiB
cd ${GRAFT_REPO_ROOT:-$PWD}
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['path_ms'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for rep in 1 2; do
  for l in "$@"; do
    export MZD_LIB=$PWD/tmp_ab/$l
    for g in 1 4; do timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 5 --workload corpus --corpus-gib $g $BENCH_EXTRA 2>/dev/null | pick "$l corpus $g GiB"; done
  done
done
