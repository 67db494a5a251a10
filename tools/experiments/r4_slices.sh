# (the MZD_EXP_SLICES / MZD_EXP_S2_LOW hooks lived in commit 8e291ae only; result: profiles/r4_seq_exec_slices.txt)
# the sequence stage in slices of whole rounds, each slice's execution beside the sequence stage of the next (MZD_EXP_SLICES),
# with fewer chains per CU (MZD_SEQ_NCH: LDS left for execution wavefronts) and the second stream at low priority (MZD_EXP_S2_LOW)
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
export MZD_LIB=$PWD/tmp_ab/libmzd_exp.so
run() { timeout 200 python bench.py --cpu-seconds 0 --no-ceiling --steps 8 "$@" 2>/dev/null; }
run | pick "shipped"
for nch in 56 52 48 44 40 32; do
  for low in "" 1; do
    MZD_EXP_SLICES=16 MZD_SEQ_NCH=$nch MZD_EXP_S2_LOW=$low run | pick "slices=16 nch=$nch low=$low"
  done
done
for sl in 2 4 8; do
  MZD_EXP_SLICES=$sl MZD_SEQ_NCH=48 run | pick "slices=$sl nch=48"
done
