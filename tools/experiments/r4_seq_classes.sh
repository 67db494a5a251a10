# real data: the sequence stage's long chains and short chains as two launches AT THE SAME TIME (two streams), the short ones with the
# LDS slot of their own tables (MZD_EXP_SEQ_CLASSES=1) against the one sorted launch
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
export MZD_LIB=$PWD/tmp_ab/libmzd_exp.so
for rep in 1 2; do
  for g in 4 2; do
    timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --workload corpus --corpus-gib $g 2>/dev/null | pick "corpus $g GiB one launch"
    MZD_EXP_SEQ_CLASSES=1 timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --workload corpus --corpus-gib $g 2>/dev/null | pick "corpus $g GiB two launches"
  done
done
