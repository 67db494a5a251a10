# where block mode starts to pay: 8 GiB / 2 GiB of output as n frames, by the library's choice (0), forced block mode (4), forced serial k_exec_c (5)
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
run() { timeout 500 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames $1 --frame-bytes $2 --gen-seconds 100 --exec-variant $3 2>/dev/null | pick "$1 x $(($2 >> 20)) MiB exec_variant $3"; }
for cfg in "2048 4194304" "512 16777216" "256 33554432" "128 67108864" "512 4194304" "128 16777216" "64 33554432"; do
  set -- $cfg
  for v in 0 4 5; do run $1 $2 $v; done
done
