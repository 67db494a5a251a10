# round 3: code-placement sensitivity -- every kernel with hand-written stretches once more with its instruction stream four bytes later
cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for lib in "" tmp_ab/libmzd_shq4.so tmp_ab/libmzd_shex.so; do
MZD_LIB=$lib timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --no-split --steps 5 --warmup 2 2>/dev/null | pick "cfg4 nosplit $lib"
done
for lib in "" tmp_ab/libmzd_shxb.so; do
MZD_LIB=$lib timeout 600 python bench.py --cpu-seconds 0 --no-ceiling --no-split --exec-variant 2 --steps 5 --warmup 2 2>/dev/null | pick "cfg4 nosplit k_exec_b $lib"
done
