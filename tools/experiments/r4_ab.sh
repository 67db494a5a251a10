# same-box A/B of two libraries on config 4 (split and alone), alternating: usage r4_ab.sh <libA under tmp_ab or "shipped"> <libB>
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for rep in 1 2; do
  for l in "$@"; do
    if [ "$l" = shipped ]; then unset MZD_LIB; else export MZD_LIB=$PWD/tmp_ab/$l; fi
    timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 8 2>/dev/null | pick "$l split"
    timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 8 --no-split 2>/dev/null | pick "$l no-split"
  done
done
