cd $GRAFT_REPO_ROOT
B="python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-ceiling --no-split"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for t in 256 192; do for l in 0 20000 26000 32000 40000 53000; do MZD_EXEC_MIN_LDS=$l $B --exec-threads $t 2>/dev/null | pick "threads=$t exec_min_lds=$l"; done; done
for l in 20000 26000 40000; do MZD_EXEC_MIN_LDS=$l $B --exec-threads 256 --exec-chunk 16384 2>/dev/null | pick "threads=256 chunk=16384 exec_min_lds=$l"; done
for l in 20000 26000 40000; do MZD_EXEC_MIN_LDS=$l $B --exec-threads 256 --exec-chunk 4096 2>/dev/null | pick "threads=256 chunk=4096 exec_min_lds=$l"; done
