# k_seq_q4: which stage bounds the step -- per-stage waits (-DMZD_Q4_PROF) and the chain wavefronts with stages B / C as no-ops (-DMZD_EXP_FAST_BC; wrong results)
cd ${GRAFT_REPO_ROOT:-$PWD}
for l in "$@"; do echo "== $l"; MZD_LIB=$PWD/tmp_ab/$l timeout 100 python3 tools/q4_stats.py 13824 2>&1 | grep -E "^B|^C|cycles/step|workgroup|kernel ms" | head -12; done
