#!/bin/bash
# Stall attribution of the convolution's compute waves: SQ counter passes over the headline bench, each set in its own
# rocprofv3 run with --kernel-trace only (MI355X_MICROARCH.md: 8 SQ slots per pass; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES).
#   gpurun --timeout 900 -- 'bash tools/pmc_stall.sh r04 f16'
set -o pipefail
TAG=${1:-r04}
PREC=${2:-f16}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --precision $PREC --no-cpu-baseline --no-parity-leg --no-train-leg --no-e2e-leg"
rocprofv3 -L > $O/counters_${TAG}.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_${TAG}_pmc_stall1 -o run -- $B > $O/prof_${TAG}_pmc_stall1.log 2>&1 && echo stall1 ok &&
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $O/prof_${TAG}_pmc_stall2 -o run -- $B > $O/prof_${TAG}_pmc_stall2.log 2>&1 && echo stall2 ok &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/prof_${TAG}_pmc_stall3 -o run -- $B > $O/prof_${TAG}_pmc_stall3.log 2>&1 && echo stall3 ok
find $O -name "*kernel_trace.csv" -size +8M -delete
