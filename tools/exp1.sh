timeout 600 python -m pytest tests/test_gpu_corpus.py -m gpu -x -q 2>&1 | tail -2
for n in 61 15104; do
MZD_LIB=$PWD/sparkzstd_amd/libmzd_prof.so timeout 200 python bench.py --frames-per-gpu $n --seq-variant 3 --steps 1 --warmup 0 --cpu-seconds 0 --gen-seconds 8 2>/dev/null | grep -v "^{" | head -8
done
for n in 61 15104 65536; do
    timeout 200 python bench.py --frames-per-gpu $n --seq-variant 3 --steps 4 --warmup 1 --cpu-seconds 0 --gen-seconds 8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('n$n', d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])"
done
