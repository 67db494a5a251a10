pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], round(d['value']), d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for n in 4096 32768 65536; do timeout 300 python bench.py --config 3 --cpu-seconds 0 --no-ceiling --steps 10 --warmup 2 --frames $n --gen-seconds 60 2>/dev/null | pick "config 3, $n frames, defaults"; done
timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 2>/dev/null | pick "config 4 defaults"
timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --workload corpus 2>/dev/null | pick "corpus defaults"
timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --frames 8192 --frame-bytes 1048576 2>/dev/null | pick "8192 x 1 MiB defaults"
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -1
python tools/experiments/r4_mixed2.py 2>&1 | grep defaults
