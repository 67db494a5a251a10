cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 4; do
rm -rf gpurun_out/r3_sj_prof
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r3_sj_prof --output-format csv -- python3 bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 1 --frame-bytes 1073741824 --gen-seconds 200 --exec-variant $v > /dev/null 2>&1
f=$(find gpurun_out/r3_sj_prof -name "*kernel_stats.csv" | head -1)
echo "exec_variant $v"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'mzd' in r['Name']: print('  ', r['Name'][:44], r['Calls'], round(float(r['AverageNs'])/1e6, 3))
PY
done
rm -rf gpurun_out/r3_sj_prof
