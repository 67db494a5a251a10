# four-byte records (k_seq_q4<true> -> k_exec_c<false, true>) and 8-byte staging loads for short far matches: the GPU tests that touch
# them on the shipped library, then same-box A/B: libraries under tmp_ab (MZD_EXP_NO_REC4=1 turns the records back to 8 bytes in an
# -DMZD_EXPERIMENTS build)
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "four_byte or oracle_trace or decodecorpus_bit_exact_on_gpu or fuzz or escape_codes or randomized or synthetic_configs or config4 or config3 or config2 or corrupt or truncat or raw_rle" 2>&1 | tail -4
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for rep in 1 2; do
  for l in "$@"; do
    export MZD_LIB=$PWD/tmp_ab/$l
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 8 --no-split 2>/dev/null | pick "$l no-split"
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 8 2>/dev/null | pick "$l split"
  done
  export MZD_LIB=$PWD/tmp_ab/libmzd_rec4.so
  MZD_EXP_NO_REC4=1 timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 8 --no-split 2>/dev/null | pick "libmzd_rec4.so with 8-byte records no-split"
done
