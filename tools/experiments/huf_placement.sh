cd ${GRAFT_REPO_ROOT:-$PWD}
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for rep in 1 2; do
  timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 8 2>/dev/null | pick "default"
  timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 8 --huf-variant 1 2>/dev/null | pick "k_huf beside the sequence stage"
  timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 8 --huf-variant 2 2>/dev/null | pick "k_huf_seg"
done
