# k_seq_q4's step: the sequence-related GPU tests on the shipped library, then same-box A/B of libraries under tmp_ab, then cycles
# per step (tools/q4_stats.py) of the -DMZD_Q4_STATS builds named in STATS_LIBS
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout 900 python3 -m pytest tests -m gpu -x -q -k "decodecorpus_bit_exact_on_gpu or oracle_trace or stage_boundaries or fuzzed_frames or escape_codes or randomized or synthetic_configs or multi_block or window_by_window or corrupt or truncat or fuzz" 2>&1 | tail -4
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for rep in 1 2; do
  for l in "$@"; do
    export MZD_LIB=$PWD/tmp_ab/$l
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 8 --no-split 2>/dev/null | pick "$l no-split"
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 8 2>/dev/null | pick "$l split"
  done
done
for l in ${STATS_LIBS:-libmzd_q4stats.so}; do echo $l; MZD_LIB=$PWD/tmp_ab/$l timeout 100 python3 tools/q4_stats.py 13824; done
# per-stage waits (-DMZD_Q4_PROF) / the chain wavefronts with stages B and C as no-ops (-DMZD_EXP_FAST_BC): PROF_LIBS="lib ..."
for l in $PROF_LIBS; do echo "== $l"; MZD_LIB=$PWD/tmp_ab/$l timeout 100 python3 tools/q4_stats.py 13824 2>&1 | grep -E "^B|^C|cycles/step|workgroup|kernel ms" | head -12; done
