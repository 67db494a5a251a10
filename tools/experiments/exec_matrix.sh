# libraries under tmp_ab over the round's other workloads (real data, multi-block frames, few large frames, small frames)
cd ${GRAFT_REPO_ROOT:-$PWD}
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
LIBS="$@"
for rep in 1 2; do
  for l in $LIBS; do
    export MZD_LIB=$PWD/tmp_ab/$l
    for g in 1 4; do timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 5 --workload corpus --corpus-gib $g 2>/dev/null | pick "$l corpus $g GiB"; done
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 5 --frames 8192 2>/dev/null | pick "$l 8192 x 128 KiB"
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 5 --frames 131072 --frame-bytes 4096 2>/dev/null | pick "$l 131072 x 4 KiB"
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames 8192 --frame-bytes 1048576 2>/dev/null | pick "$l 8192 x 1 MiB"
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames 1 --frame-bytes 1073741824 --gen-seconds 200 2>/dev/null | pick "$l 1 x 1 GiB"
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames 16 --frame-bytes 134217728 --gen-seconds 200 2>/dev/null | pick "$l 16 x 128 MiB"
  done
done
