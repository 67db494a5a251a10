#!/bin/bash
# GPU box: rocprofv3 evidence for ONE bench configuration, all in the same gpurun:
#   1. kernel trace + stats of `bench.py --config N`            -> gpurun_out/<tag>_cfg<N>_kernel_stats.csv
#   2. FETCH_SIZE and WRITE_SIZE in two separate --pmc passes    -> gpurun_out/<tag>_traffic_cfg<N>.json
#      (stamped with the hash of the device sources and the workload, tools/summarize_profile.py)
#   3. the bench line itself, quoting that traffic file          -> gpurun_out/<tag>_cfg<N>_bench.json
# Copy what is to be kept into profiles/ (gpurun_out/ is scratch).
# usage: tools/profile_round.sh <tag> <config> [extra bench flags]
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r6}
CFG=${2:-4}
shift $(( $# < 2 ? $# : 2 ))
EXTRA="$@"
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
# (--no-verify: the profiled process launches exactly steps + warmup = 7 passes; the verification pass -- an eighth launch of every
# kernel -- belongs to the bench run at the end, which is not profiled)
B="python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --cpu-seconds 0 --no-ceiling --no-verify $EXTRA"
for k in stats fetch write; do rm -rf $R/gpurun_out/${TAG}_cfg${CFG}_$k; done
LOG=$R/gpurun_out/${TAG}_cfg${CFG}_rocprof.log
: > $LOG
run() {  # one rocprofv3 pass; a failed pass stops the script instead of leaving stale directories to be summarised
  "$@" >> $LOG 2>&1 || { echo "rocprofv3 pass failed (rc $?): $*" >&2; tail -5 $LOG >&2; exit 1; }
}
run timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_cfg${CFG}_stats --output-format csv -- $B
run timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_cfg${CFG}_fetch --output-format csv -- $B
run timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_cfg${CFG}_write --output-format csv -- $B
cd $R && python3 tools/summarize_profile.py $TAG $CFG 7 "$B"
cd $R && timeout 900 python3 bench.py --config $CFG $EXTRA --traffic-from gpurun_out/${TAG}_traffic_cfg${CFG}.json \
    > gpurun_out/${TAG}_cfg${CFG}_bench.json 2> gpurun_out/${TAG}_cfg${CFG}_bench.err
tail -1 gpurun_out/${TAG}_cfg${CFG}_bench.json
