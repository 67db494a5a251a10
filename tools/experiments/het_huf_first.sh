# real data: the Huffman stage of a heterogeneous batch beside the sequence stage (the default) or in front of it (MZD_EXP_HET_HUF_FIRST: the
# stage's own duration shows).  Library: tmp_ab/libmzd_exp.so, built with -DMZD_EXPERIMENTS.  (profiles/r5_het_huf.txt also holds the runs with
# the long streams / whole table classes through k_huf_seg, whose hooks are gone: the first became the rule, the second was far worse.)
cd ${GRAFT_REPO_ROOT:-$PWD}
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
export MZD_LIB=$PWD/tmp_ab/libmzd_exp.so
for rep in 1 2; do
  for g in 1 4; do
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 5 --workload corpus --corpus-gib $g 2>/dev/null | pick "corpus $g GiB, Huffman beside the sequence stage"
    MZD_EXP_HET_HUF_FIRST=1 timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 5 --workload corpus --corpus-gib $g 2>/dev/null | pick "corpus $g GiB, Huffman first"
  done
done
