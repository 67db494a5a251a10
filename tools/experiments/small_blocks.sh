# the 4 KiB-ring executor's block-start path: libraries under tmp_ab on the workloads that take it (real data at 1 and 4 GiB, small
# frames), after the execution-related parity tests with each
cd ${GRAFT_REPO_ROOT:-$PWD}
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
LIBS="$@"
for l in $LIBS; do
  echo "== tests with $l"
  MZD_LIB=$PWD/tmp_ab/$l timeout 1500 python3 -m pytest tests -m gpu -x -q -k "corpus or exec or stage or block or fuzz or ragged or multi" 2>&1 | tail -2
done
for rep in 1 2; do
  for l in $LIBS; do
    export MZD_LIB=$PWD/tmp_ab/$l
    for g in 1 4; do timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 5 --workload corpus --corpus-gib $g 2>/dev/null | pick "$l corpus $g GiB"; done
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 5 --frames 131072 --frame-bytes 4096 2>/dev/null | pick "$l 131072 x 4 KiB"
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 5 --frames 65536 --frame-bytes 16384 2>/dev/null | pick "$l 65536 x 16 KiB"
  done
done
