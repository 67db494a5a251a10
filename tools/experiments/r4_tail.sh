# (MZD_EXP_TAIL_FULL lived in commit 70d2693's working tree; 'packed' is what the library does now; result: profiles/r4_seq_tail_packed.txt)
# the last, partial round of the sequence stage: its chains spread over all CUs (32 per workgroup, the execution stage's head beside
# them on every CU) against packed into as few CUs as hold them (56 per workgroup, the other CUs free for the execution stage)
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
export MZD_LIB=$PWD/tmp_ab/libmzd_exp.so
for rep in 1 2; do
  timeout 200 python bench.py --cpu-seconds 0 --no-ceiling --steps 8 2>/dev/null | pick "spread"
  MZD_EXP_TAIL_FULL=1 timeout 200 python bench.py --cpu-seconds 0 --no-ceiling --steps 8 2>/dev/null | pick "packed"
done
for n in 8192 16384 32768; do
  timeout 200 python bench.py --cpu-seconds 0 --no-ceiling --steps 8 --frames $n 2>/dev/null | pick "spread frames=$n"
  MZD_EXP_TAIL_FULL=1 timeout 200 python bench.py --cpu-seconds 0 --no-ceiling --steps 8 --frames $n 2>/dev/null | pick "packed frames=$n"
done
