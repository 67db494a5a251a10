# BASELINE config 3 (Huffman-only frames) by batch size: the library's choice of Huffman kernel (0) against forced k_huf (1), k_huf_seg (2), k_huf first (3)
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], round(d['value']), d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for n in 256 1024 4096 16384 65536; do
  for v in 0 1 2 3; do
    timeout 300 python bench.py --config 3 --cpu-seconds 0 --no-ceiling --steps 10 --warmup 2 --frames $n --huf-variant $v --gen-seconds 60 2>/dev/null | pick "config 3, $n frames, huf_variant $v"
  done
done
