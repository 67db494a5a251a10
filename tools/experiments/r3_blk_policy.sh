# round 3: does mzd_batch_run's model pick the faster of {frames as serial jobs, block mode}?  8 GiB of output as n frames of s MiB
cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms']['k_exec'], d['bit_exact'])" "$1"; }
for c in "2048 4194304" "512 16777216" "256 33554432" "128 67108864"; do set -- $c
for v in 0 2 3; do timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames $1 --frame-bytes $2 --gen-seconds 200 --exec-variant $v 2>/dev/null | pick "$1 x $(($2 >> 20)) MiB exec_variant $v"; done; done
