#!/bin/bash
# Same-box A/B of gn_apply16c builds / switches (run through gpurun from the repo root):
#   base = stedm_amd/libstedm_hip_base.so (the previous build, via STEDM_HIP_LIB), new = the in-tree build, STEDM_GN_U=4 = 64 B per lane in flight.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
B="python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-parity-leg --no-train-leg --no-e2e-leg"
{
  echo "== tests (U=2 default)";  timeout -k 10 300 python3 -m pytest tests -m gpu -q -x -k "gn or GroupNorm or groupnorm" 2>&1 | tail -3 || exit 1
  echo "== tests (U=4)";  STEDM_GN_U=4 timeout -k 10 300 python3 -m pytest tests -m gpu -q -x -k "gn or GroupNorm or groupnorm or unet_vs_reference" 2>&1 | tail -3 || exit 1
  if [ -f stedm_amd/libstedm_hip_base.so ]; then echo "== bench_gn base"; STEDM_HIP_LIB=$R/stedm_amd/libstedm_hip_base.so python3 tools/bench_gn.py; fi
  echo "== bench_gn new U=2"; python3 tools/bench_gn.py
  echo "== bench_gn new U=4"; STEDM_GN_U=4 python3 tools/bench_gn.py
  echo "== bench_gn new U=4 slab 64K"; STEDM_GN_U=4 STEDM_GN_SLAB_KB=64 python3 tools/bench_gn.py
  echo "== bench_gn new U=4 slab 256K"; STEDM_GN_U=4 STEDM_GN_SLAB_KB=256 python3 tools/bench_gn.py
  for rep in 1 2; do
    if [ -f stedm_amd/libstedm_hip_base.so ]; then echo "== step base #$rep"; STEDM_HIP_LIB=$R/stedm_amd/libstedm_hip_base.so $B | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; fi
    echo "== step new U=2 #$rep"; $B | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
    echo "== step new U=4 #$rep"; STEDM_GN_U=4 $B | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
} > $O/ab_gn.log 2>&1
