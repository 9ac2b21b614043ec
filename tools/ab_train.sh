#!/bin/bash
# Same-box A/B of a training-step switch (run through gpurun from the repo root):  bash tools/ab_train.sh STEDM_WGRAD1X1_GEMM=1
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
V=${1:-STEDM_WGRAD1X1_GEMM=1}
cd $R
for rep in 1 2 3; do
  echo "== default #$rep"; python3 tools/bench_train.py --steps 10 2>&1 | grep "train step"
  echo "== $V #$rep"; env $V python3 tools/bench_train.py --steps 10 2>&1 | grep "train step"
done
