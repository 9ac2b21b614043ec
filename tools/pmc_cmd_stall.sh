#!/bin/bash
# Stall attribution (three SQ counter passes, each its own rocprofv3 run) of any python tool of this repo:
#   gpurun --timeout 600 -- 'bash tools/pmc_cmd_stall.sh r05_attn tools/bench_attn.py f16'
# -> gpurun_out/prof_<tag>_stall{1,2,3}; report: python tools/pmc_stall_report.py gpurun_out/prof_<tag>_stall '<kernel regex>'
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
SCRIPT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_${TAG}_stall1 -o run -- python3 $SCRIPT "$@" > $O/prof_${TAG}_stall1.log 2>&1 && echo stall1 ok &&
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $O/prof_${TAG}_stall2 -o run -- python3 $SCRIPT "$@" > $O/prof_${TAG}_stall2.log 2>&1 && echo stall2 ok &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/prof_${TAG}_stall3 -o run -- python3 $SCRIPT "$@" > $O/prof_${TAG}_stall3.log 2>&1 && echo stall3 ok
find $O -name "*kernel_trace.csv" -size +8M -delete
