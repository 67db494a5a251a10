# k_seq_q4 with chain slots sized to the batch's largest tables against fixed 1280-cell slots (tmp_ab/libmzd_fixedslot.so =
# the library one commit earlier) on batches of small frames, whose sequence tables are small
cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for fb in 4096 16384; do
  n=$((2147483648 / fb))
  timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --frame-bytes $fb --frames-per-gpu $n 2>/dev/null | pick "frames of $fb B x $n: sized slots"
  MZD_LIB=$PWD/tmp_ab/libmzd_fixedslot.so timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --frame-bytes $fb --frames-per-gpu $n 2>/dev/null | pick "frames of $fb B x $n: fixed slots"
done
timeout 100 python bench.py --cpu-seconds 0 --no-ceiling --steps 10 2>/dev/null | pick "config 4"
