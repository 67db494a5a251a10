# heterogeneous batches in two groups of frames (the frames that hold the longest chains on a third stream): the GPU suite with the
# candidate library, then real data at 1 / 2 / 4 GiB and the corpus's 100 frames alone, A/B of the libraries under tmp_ab
cd ${GRAFT_REPO_ROOT:-$PWD}
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
MZD_LIB=$PWD/tmp_ab/$2 timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
  for l in "$@"; do
    export MZD_LIB=$PWD/tmp_ab/$l
    for g in 1 2 4; do timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 5 --workload corpus --corpus-gib $g 2>/dev/null | pick "$l corpus $g GiB"; done
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 5 --workload corpus --corpus-gib 0.25 2>/dev/null | pick "$l corpus 0.25 GiB"
  done
done
