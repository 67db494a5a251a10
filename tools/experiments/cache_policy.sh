# k_exec_c with `nt` on its read-once streams (records / literals / staged match sources), frames in flight per CU, and the L2 counters
# of each: usage r5_nt.sh <lib under tmp_ab> ...   (libraries built with -DMZD_EXPERIMENTS -DMZD_XC_NT=<mask>)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
ctr() {  # <label> <counters...>: per-pass sums for k_exec_c over 3 + 1 launched passes
  local label=$1; shift
  rm -rf /tmp/r5ctr; (cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/r5ctr --output-format csv -- python3 $R/bench.py --cpu-seconds 0 --no-ceiling --no-split --no-verify --steps 3 --warmup 1 >/dev/null 2>&1)
  python3 - "$label" <<'PY'
import csv, glob, collections, sys, re
fs = glob.glob('/tmp/r5ctr/*/*_counter_collection.csv')
agg = collections.defaultdict(collections.Counter); calls = collections.Counter()
for r in csv.DictReader(open(fs[0])) if fs else []:
    m = re.search(r"mzd::(k_\w+)", r["Kernel_Name"])
    if m: agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
for k in ("k_exec_c", "k_seq_q4", "k_huf"):
    if k in agg: print(sys.argv[1], k, {c: round(v / 4) for c, v in agg[k].items()})
PY
}
for rep in 1 2; do
  for l in "$@"; do
    export MZD_LIB=$R/tmp_ab/$l
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 8 --no-split 2>/dev/null | pick "$l no-split"
    timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 8 2>/dev/null | pick "$l split"
  done
done
for l in ${CTR_LIBS:-$@}; do
  export MZD_LIB=$R/tmp_ab/$l
  ctr "$l" FETCH_SIZE
  ctr "$l" WRITE_SIZE
  ctr "$l" TCC_HIT_sum TCC_MISS_sum
  ctr "$l" TCC_REQ_sum TCC_READ_sum
done
# frames in flight per CU (extra dynamic LDS per frame): 20 (default) / 18 / 16 / 14 / 12
for l in ${SWEEP_LIBS:-$@}; do
  export MZD_LIB=$R/tmp_ab/$l
  for x in 0 1536 2816 4352 6144; do
    MZD_EXP_XC_LDS=$x timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 6 --no-split 2>/dev/null | pick "$l extra-lds $x no-split"
  done
done
