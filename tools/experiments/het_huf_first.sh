# real data: the Huffman stage of a heterogeneous batch -- beside the sequence stage (the default) or in front of it (MZD_EXP_HET_HUF_FIRST: the
# kernel's own duration), and with the quads of every table class whose longest stream has 4 / 8 / 16 / 32 KiB or more through k_huf_seg
# (MZD_EXP_HUF_LONG = 0..3) instead of a lane each.  Library: tmp_ab/libmzd_exp.so, built with -DMZD_EXPERIMENTS.
cd ${GRAFT_REPO_ROOT:-$PWD}
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
export MZD_LIB=$PWD/tmp_ab/libmzd_exp.so
for rep in 1 2; do
  for g in 1 4; do
    for k in off 0 1 2 3; do
      [ $k = off ] && unset MZD_EXP_HUF_LONG || export MZD_EXP_HUF_LONG=$k
      timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 5 --workload corpus --corpus-gib $g 2>/dev/null | pick "corpus $g GiB, beside, long streams $k"
      MZD_EXP_HET_HUF_FIRST=1 timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 5 --workload corpus --corpus-gib $g 2>/dev/null | pick "corpus $g GiB, first,  long streams $k"
    done
  done
done
