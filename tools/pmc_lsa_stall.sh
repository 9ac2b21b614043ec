#!/bin/bash
# Stall attribution of the LSA attention kernels (bf16 and MX-fp8): SQ counter passes over tools/bench_lsa.py, each set in its own run.
#   gpurun --timeout 600 -- 'bash tools/pmc_lsa_stall.sh r04'
set -o pipefail
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for M in bf16 fp8; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_${TAG}_lsa_${M}_stall1 -o run -- python3 $R/tools/bench_lsa.py $M 64 > $O/prof_${TAG}_lsa_${M}_stall1.log 2>&1 && echo $M stall1 ok &&
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $O/prof_${TAG}_lsa_${M}_stall2 -o run -- python3 $R/tools/bench_lsa.py $M 64 > $O/prof_${TAG}_lsa_${M}_stall2.log 2>&1 && echo $M stall2 ok &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/prof_${TAG}_lsa_${M}_stall3 -o run -- python3 $R/tools/bench_lsa.py $M 64 > $O/prof_${TAG}_lsa_${M}_stall3.log 2>&1 && echo $M stall3 ok
done
find $O -name "*kernel_trace.csv" -size +8M -delete
