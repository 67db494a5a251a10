# block mode: its parity tests on the shipped library, then same-box A/B of libraries under tmp_ab on few large frames
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "block_mode or blocks_ or multi_block or large" 2>&1 | tail -4
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
LIBS="$@"
for rep in 1 2; do
  for l in $LIBS; do
    export MZD_LIB=$PWD/tmp_ab/$l
    for cfg in "1 1073741824" "1 268435456" "4 268435456" "16 134217728"; do set -- $cfg; timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames $1 --frame-bytes $2 --gen-seconds 200 2>/dev/null | pick "$l $1 x $(($2 >> 20)) MiB"; done
  done
done
