# k_huf: the Huffman-related GPU tests on the shipped library, then same-box A/B of libraries under tmp_ab (config 4, and the 16 384-frame shard where k_huf runs first too)
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout 1200 python3 -m pytest tests -m gpu -x -q -k "huf or literal or oracle_trace or stage_boundaries or decodecorpus_bit_exact_on_gpu or fuzz or corrupt or truncat or config3 or config4 or synthetic" 2>&1 | tail -4
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for rep in 1 2; do
  for l in "$@"; do
    export MZD_LIB=$PWD/tmp_ab/$l
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 8 2>/dev/null | pick "$l config 4"
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 8 --frames 16384 2>/dev/null | pick "$l 16384 frames"
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 8 --frames 8192 2>/dev/null | pick "$l 8192 frames"
  done
done
