#!/bin/bash
# GPU box: SQ / TA / LDS counters of the default bench workload, one --pmc pass per counter group
# (kernel-trace only).  Outputs under gpurun_out/<tag>_pmc{A,B,C}; tools/summarize_counters.py turns
# them into profiles/<tag>_sq_counters.csv.
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r1_v5}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0"
rm -rf $R/gpurun_out/${TAG}_pmcA $R/gpurun_out/${TAG}_pmcB $R/gpurun_out/${TAG}_pmcC
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES -d $R/gpurun_out/${TAG}_pmcA --output-format csv -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/${TAG}_pmcB --output-format csv -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_pmcC --output-format csv -- $B > /dev/null 2>&1
ls $R/gpurun_out/${TAG}_pmcA/*/ | head -3
