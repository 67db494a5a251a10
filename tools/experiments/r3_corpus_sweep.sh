# round 3: corpus workload over the residency caps of the Huffman kernel (LDS request per wavefront) and of k_exec_b (extra LDS per frame)
cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for h in 8192 16384 24576 32768 65536; do timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --workload corpus --steps 4 --warmup 1 --huf-min-lds $h 2>/dev/null | pick "huf_min_lds $h"; done
for c in 1024 2560 5120; do timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --workload corpus --steps 4 --warmup 1 --exec-chunk $c 2>/dev/null | pick "exec_chunk $c"; done
timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --workload corpus --steps 4 --warmup 1 --huf-variant 2 2>/dev/null | pick "huf_variant 2"
