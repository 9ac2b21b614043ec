#!/bin/bash
# Collects the round's profile evidence on the GPU box (run through gpurun from the repo root): the rocprofv3 kernel-trace summaries of the
# denoising bench, the training step, the two style encoders, and the three PMC passes (each counter set in its own run, kernel trace only —
# MI355X_MICROARCH.md). Outputs under gpurun_out/prof_<tag>_*; tools/pmc_report.py + the copy into profiles/ happen afterwards.
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r03'
set -o pipefail
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-leg --no-train-leg --no-e2e-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_step -o run -- $B > $O/prof_${TAG}_step.log 2>&1 && echo step ok &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_parity -o run -- $B --precision parity > $O/prof_${TAG}_parity.log 2>&1 && echo parity ok &&
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_${TAG}_pmc_mfma_parity -o run -- $B --precision parity > $O/prof_${TAG}_pmc_mfma_parity.log 2>&1 && echo mfma parity ok &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_train -o run -- python3 $R/tools/bench_train.py > $O/prof_${TAG}_train.log 2>&1 && echo train ok &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_svit -o run -- python3 $R/tools/bench_svit.py bf16 64 > $O/prof_${TAG}_svit.log 2>&1 && echo svit ok &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_svit_fp8 -o run -- python3 $R/tools/bench_svit.py fp8 64 > $O/prof_${TAG}_svit_fp8.log 2>&1 && echo svit fp8 ok &&
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU --output-format csv -d $O/prof_${TAG}_pmc_lsa -o run -- python3 $R/tools/bench_lsa.py bf16 64 > $O/prof_${TAG}_pmc_lsa.log 2>&1 && echo lsa pmc ok &&
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU --output-format csv -d $O/prof_${TAG}_pmc_lsa_fp8 -o run -- python3 $R/tools/bench_lsa.py fp8 64 > $O/prof_${TAG}_pmc_lsa_fp8.log 2>&1 && echo lsa fp8 pmc ok &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_swin -o run -- python3 $R/tools/bench_swin.py bf16 128 32 > $O/prof_${TAG}_swin.log 2>&1 && echo swin ok &&
# the three PMC passes of the headline binary run LAST: no kernel source changes after them (roofline.traffic is keyed to the conv sources)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_${TAG}_pmc_mfma -o run -- $B > $O/prof_${TAG}_pmc_mfma.log 2>&1 && echo mfma ok &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_${TAG}_pmc_fetch -o run -- $B > $O/prof_${TAG}_pmc_fetch.log 2>&1 && echo fetch ok &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_${TAG}_pmc_write -o run -- $B > $O/prof_${TAG}_pmc_write.log 2>&1 && echo write ok
# the trace CSVs of the PMC passes are large: keep the counter files and the stats only
find $O -name "*kernel_trace.csv" -size +8M -delete
