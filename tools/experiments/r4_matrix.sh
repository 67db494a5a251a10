# k_exec (1) / k_exec_b (2) / k_exec_c (5) over the round's workloads (whole pass, default stream plan)
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for v in 1 2 5; do
  for n in 32768 16384 8192; do
    timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 8 --frames $n --exec-variant $v 2>/dev/null | pick "v$v frames=$n"
  done
  timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 4 --frames 8192 --frame-bytes 1048576 --exec-variant $v 2>/dev/null | pick "v$v 8192 x 1 MiB"
  timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --frames 131072 --frame-bytes 4096 --exec-variant $v 2>/dev/null | pick "v$v 131072 x 4 KiB"
  timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --steps 4 --workload corpus --exec-variant $v 2>/dev/null | pick "v$v corpus"
done
