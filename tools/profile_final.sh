#!/bin/bash
# End-of-round refresh on the final binary (run through gpurun from the repo root): kernel-trace summaries of the denoising step and the
# training step, then the full default bench line. Outputs under gpurun_out/; tools/stats_md.py turns the summaries into profiles/.
set -o pipefail
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-leg --no-train-leg --no-e2e-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_step -o run -- $B > $O/prof_${TAG}_step.log 2>&1 && echo step ok &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_train -o run -- python3 $R/tools/bench_train.py > $O/prof_${TAG}_train.log 2>&1 && echo train ok &&
cd $R && timeout -k 10 400 python3 bench.py > $O/bench_final.json 2> $O/bench_final.err && echo bench ok
find $O -name "*kernel_trace.csv" -size +8M -delete
