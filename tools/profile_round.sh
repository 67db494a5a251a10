#!/bin/bash
# GPU box: tests, default bench, rocprofv3 kernel stats and HBM-traffic counters of the same command.
# Outputs go to gpurun_out/ (scratch); copy what is to be kept into profiles/.
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r1_v5}
cd $R && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd $R && timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -1 gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0"
rm -rf $R/gpurun_out/${TAG}_stats $R/gpurun_out/${TAG}_fetch $R/gpurun_out/${TAG}_write
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats --output-format csv -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch --output-format csv -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write --output-format csv -- $B > /dev/null 2>&1
ls $R/gpurun_out/${TAG}_stats/*/ | head
