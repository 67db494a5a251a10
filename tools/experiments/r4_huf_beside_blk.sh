# block mode: the Huffman kernel beside the sequence stage (second stream) instead of in front of it (tmp_ab/libmzd_hufbeside.so: one line), same box
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for rep in 1 2; do
 for lib in "" $PWD/tmp_ab/libmzd_hufbeside.so; do
  if [ -z "$lib" ]; then unset MZD_LIB; tag=first; else export MZD_LIB=$lib; tag=beside; fi
  for g in 0.01 0.1; do timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 8 --workload corpus --corpus-gib $g 2>/dev/null | pick "corpus $g GiB huf $tag"; done
  timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames 1 --frame-bytes 1073741824 --gen-seconds 200 2>/dev/null | pick "1 x 1 GiB huf $tag"
  timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames 1 --frame-bytes 67108864 2>/dev/null | pick "1 x 64 MiB huf $tag"
  timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 64 --frame-bytes 134217728 --gen-seconds 200 2>/dev/null | pick "64 x 128 MiB huf $tag"
 done
done
