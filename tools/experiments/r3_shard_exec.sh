# round 3: the 8 192-frame shard of configs[4] over the execution kernel's geometry (32 frames per CU: all resident at once?)
cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for args in "" "--exec-variant 2" "--exec-variant 1 --exec-chunk 4096 --exec-threads 64" "--exec-variant 1 --exec-chunk 4096 --exec-threads 128" "--exec-variant 1 --exec-threads 64" "--exec-variant 1 --exec-threads 256"; do
timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --frames 8192 $args 2>/dev/null | pick "8192 [$args]"; done
