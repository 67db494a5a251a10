cd $GRAFT_REPO_ROOT
B="python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-ceiling --no-split"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for l in 0 10000 13000 16000 20000 26000 40000 80000; do MZD_EXEC_MIN_LDS=$l $B 2>/dev/null | pick "exec_min_lds=$l"; done
for l in 4096; do for m in 0 8000 10000 16000 26000; do MZD_EXEC_MIN_LDS=$m $B --exec-chunk $l 2>/dev/null | pick "chunk=$l exec_min_lds=$m"; done; done
for n in 54 48 44 40 36; do MZD_SEQ_NCH=$n $B 2>/dev/null | pick "nch=$n"; done
