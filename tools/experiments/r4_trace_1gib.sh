cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/tr1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/tr1 --output-format csv -- python3 bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --workload corpus --corpus-gib 1 > /dev/null 2>&1
f=$(find gpurun_out/tr1 -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.reader(open(sys.argv[1])))[1:9]: print(r[0][:60], r[1], round(float(r[3])/1e6,3))
PY
rm -rf gpurun_out/tr1
