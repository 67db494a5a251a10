# which Huffman kernel for batches WITH sequences and 16 k - 128 k streams: libraries under tmp_ab over the shard sizes of configs[4] and few large frames
cd ${GRAFT_REPO_ROOT:-$PWD}
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for rep in 1 2; do
  for l in "$@"; do
    export MZD_LIB=$PWD/tmp_ab/$l
    for n in 4096 8192 16384 32768; do timeout 300 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 10 --frames $n 2>/dev/null | pick "$l $n frames"; done
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames 1 --frame-bytes 268435456 --gen-seconds 200 2>/dev/null | pick "$l 1 x 256 MiB"
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames 16 --frame-bytes 16777216 --gen-seconds 200 2>/dev/null | pick "$l 16 x 16 MiB"
  done
done
