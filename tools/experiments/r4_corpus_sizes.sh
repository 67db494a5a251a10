# real data: the pass by batch size (replicas of the 100 decodecorpus frames), execution kernels 0 (k_exec_c) / 2 (k_exec_b) / 1 (k_exec)
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['frames'])" "$1"; }
for g in 0.01 0.1 0.25 0.5 1 1.5 2 4; do timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --workload corpus --corpus-gib $g 2>/dev/null | pick "corpus $g GiB"; done
for v in 2; do for g in 0.25 1; do timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --workload corpus --corpus-gib $g --exec-variant $v 2>/dev/null | pick "corpus $g GiB exec_variant $v"; done; done
