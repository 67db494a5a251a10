cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "1 1073741824" "64 134217728"; do set -- $spec
rm -rf gpurun_out/r3_sj_prof
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r3_sj_prof --output-format csv -- python3 bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames $1 --frame-bytes $2 --gen-seconds 200 > /dev/null 2>&1
f=$(find gpurun_out/r3_sj_prof -name "*kernel_stats.csv" | head -1)
echo "$1 x $2"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'mzd' in r['Name'] and float(r['AverageNs']) > 5e4: print('  ', r['Name'][:44], r['Calls'], round(float(r['AverageNs'])/1e6, 3))
PY
done
rm -rf gpurun_out/r3_sj_prof
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 64 --frame-bytes 134217728 --gen-seconds 200 --exec-variant 1 2>/dev/null | pick "64 x 128 MiB serial (k_exec)"
timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 1 --warmup 1 --frames 1 --frame-bytes 1073741824 --gen-seconds 200 --exec-variant 1 2>/dev/null | pick "1 x 1 GiB serial (k_exec)"
