# k_huf's wavefronts per CU (config 4: 4 096 wavefronts, 16.4 KB of LDS each = 10 per CU = 1.6 rounds): residency caps through --huf-min-lds
cd ${GRAFT_REPO_ROOT:-$PWD}
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for rep in 1 2; do
  for lds in 0 18432 20480 23404 27306 32768; do timeout 300 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 8 --huf-min-lds $lds 2>/dev/null | pick "huf_min_lds $lds"; done
done
