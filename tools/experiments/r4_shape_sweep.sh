# the default path over batch shapes: anything that is not monotone in the batch size is a policy decision to look at
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], round(d['value']), d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for fb in 4096 32768 131072 524288 2097152; do
  for n in 64 256 1024 4096 16384; do
    [ $((fb * n)) -gt 8589934592 ] && continue
    timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 5 --warmup 1 --frames $n --frame-bytes $fb --gen-seconds 60 2>/dev/null | pick "$n x $((fb >> 10)) KiB"
  done
done
