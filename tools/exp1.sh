timeout 600 python -m pytest tests/test_gpu_corpus.py -m gpu -x -q 2>&1 | tail -3
for n in 61 15104 65536; do
  for var in 3; do
    timeout 200 python bench.py --frames-per-gpu $n --seq-variant $var --steps 4 --warmup 1 --cpu-seconds 0 --gen-seconds 8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('n$n var$var', d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])"
  done
done
