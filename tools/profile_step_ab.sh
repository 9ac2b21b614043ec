#!/bin/bash
# A/B kernel-trace summaries of the headline step: gpurun -- 'bash tools/profile_step_ab.sh <tag> [ENV=1 ...]' -> gpurun_out/prof_<tag>_step{A,B}
# (A: as shipped; B: with the given environment assignments), then tools/stats_md.py on each.
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-leg --no-train-leg --no-e2e-leg --ref128-batch 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_stepA -o run -- $B > $O/prof_${TAG}_stepA.log 2>&1 && echo A ok
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_stepB -o run -- $B > $O/prof_${TAG}_stepB.log 2>&1 && echo B ok
cd $R
python3 tools/stats_md.py gpurun_out/prof_${TAG}_stepA ${TAG}_stepA "A" "$B" 4 "denoising step" ddim_step 5 | tail -1
python3 tools/stats_md.py gpurun_out/prof_${TAG}_stepB ${TAG}_stepB "B: $*" "$B" 4 "denoising step" ddim_step 5 | tail -1
