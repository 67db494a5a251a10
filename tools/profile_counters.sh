#!/bin/bash
# GPU box: SQ / TA / LDS counters of one bench configuration, one --pmc pass per counter group (kernel-trace only).
# Outputs under gpurun_out/<tag>_cfg<N>_pmc{A,B,C}; tools/summarize_counters.py <tag> <cfg> <steps> prints / writes the table.
# usage: tools/profile_counters.sh <tag> <config> [extra bench flags]
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r6}
CFG=${2:-4}
shift $(( $# < 2 ? $# : 2 ))
EXTRA="$@"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --cpu-seconds 0 --no-ceiling --no-verify $EXTRA"  # (4 passes launched, none for verification)
LOG=$R/gpurun_out/${TAG}_cfg${CFG}_rocprof_pmc.log; mkdir -p $R/gpurun_out; : > $LOG
for g in A B C D; do rm -rf $R/gpurun_out/${TAG}_cfg${CFG}_pmc$g; done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES -d $R/gpurun_out/${TAG}_cfg${CFG}_pmcA --output-format csv -- $B >> $LOG 2>&1 || { echo "rocprofv3 counter pass failed" >&2; tail -5 $LOG >&2; exit 1; }
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/${TAG}_cfg${CFG}_pmcB --output-format csv -- $B >> $LOG 2>&1 || { echo "rocprofv3 counter pass failed" >&2; tail -5 $LOG >&2; exit 1; }
timeout 600 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_cfg${CFG}_pmcC --output-format csv -- $B >> $LOG 2>&1 || { echo "rocprofv3 counter pass failed" >&2; tail -5 $LOG >&2; exit 1; }
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $R/gpurun_out/${TAG}_cfg${CFG}_pmcD --output-format csv -- $B >> $LOG 2>&1 || { echo "rocprofv3 counter pass failed" >&2; tail -5 $LOG >&2; exit 1; }
cd $R && python3 tools/summarize_counters.py $TAG $CFG 4 "$B"
