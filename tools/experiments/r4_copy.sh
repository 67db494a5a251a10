# config 2 (Raw / RLE frames): k_copy_blocks with chunks of 8 / 16 (shipped) / 32 / 64 KiB per workgroup, same box
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['ms_per_step'], r['kernel_ms'].get('k_exec'), r.get('frac_of_copy_ceiling'), d['bit_exact'])" "$1"; }
for rep in 1 2; do
  for c in 16384 8192 32768 65536; do
    if [ $c = 16384 ]; then unset MZD_LIB; else export MZD_LIB=$PWD/tmp_ab/libmzd_copy$c.so; fi
    timeout 200 python bench.py --config 2 --cpu-seconds 0 --steps 50 --warmup 5 2>/dev/null | pick "chunk $c"
  done
done
