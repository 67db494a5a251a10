# k_seq_q4's hot loop at every dword offset inside its 64-byte instruction line (-DMZD_Q4_PADW=0..15): k_seq alone (--no-split), twice
cd ${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
  for n in 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15; do
    MZD_LIB=$PWD/tmp_ab/libmzd_pad$n.so timeout 300 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 6 --no-split 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('padw', sys.argv[1], 'k_seq', d['roofline']['kernel_ms']['k_seq'], 'pass', d['ms_per_step'], d['bit_exact'])" $n
  done
done
