cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r3_het_trace
timeout 900 rocprofv3 --kernel-trace -d gpurun_out/r3_het_trace --output-format csv -- python3 bench.py --cpu-seconds 0 --no-ceiling --workload corpus --steps 2 --warmup 1 > /dev/null 2>&1
f=$(find gpurun_out/r3_het_trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'mzd::k_' in r['Kernel_Name']]
rows = rows[-8:]
t0 = min(int(r['Start_Timestamp']) for r in rows)
for r in rows:
    print(r['Kernel_Name'][:28], round((int(r['Start_Timestamp']) - t0) / 1e6, 3), round((int(r['End_Timestamp']) - t0) / 1e6, 3), r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Stream_Id') or r.get('Queue_Id'))
PY
rm -rf gpurun_out/r3_het_trace
