# real data: k_exec_c's wavefronts of LARGE frames at a higher issue priority (s_setprio by frame size; -DMZD_XC_PRIO, out of tree) -- the
# largest frame's serial walk is the execution stage's critical path (6.9 ms for 2 300 frames on an empty chip, 8.9 for 37 200)
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for rep in 1 2; do
  for lib in "" $PWD/tmp_ab/libmzd_prio.so; do
    if [ -z "$lib" ]; then unset MZD_LIB; tag=shipped; else export MZD_LIB=$lib; tag=prio; fi
    for g in 4 1 0.25; do timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --workload corpus --corpus-gib $g 2>/dev/null | pick "corpus $g GiB $tag"; done
  done
done
