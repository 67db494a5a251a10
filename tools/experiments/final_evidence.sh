cd ${GRAFT_REPO_ROOT:-$PWD}
bash tools/experiments/full_gpu_check.sh 300000 > gpurun_out/r6_full.txt 2>&1
bash tools/experiments/round_profiles.sh r6 > gpurun_out/r6_round.txt 2>&1
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'], d.get('library_chose'))" "$1"; }
for g in 0.25 1 2 3 4; do timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 5 --workload corpus --corpus-gib $g 2>/dev/null | pick "corpus $g GiB"; done > gpurun_out/r6_corpus_sizes.txt 2>&1
