# config 2 (Raw / RLE blocks only: the pass is k_copy_blocks): same-box A/B of libraries under tmp_ab, with the copy ceiling measured beside it
cd ${GRAFT_REPO_ROOT:-$PWD}
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['ms_per_step'], r['kernel_ms'], 'ceiling', r['copy_ceiling'], 'frac of it', r['frac_of_copy_ceiling'], d['bit_exact'])" "$1"; }
for rep in 1 2 3; do
  for l in "$@"; do
    export MZD_LIB=$PWD/tmp_ab/$l
    timeout 300 python3 bench.py --config 2 --cpu-seconds 0 --steps 50 --warmup 5 2>/dev/null | pick "$l config 2"
  done
done
