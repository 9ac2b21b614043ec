#!/bin/bash
# Builds tools/_ab/lib_attn_diag<bits>.so for bits in "$@" (default: 1 2 4 8 3): the shipped objects with attn.hip recompiled under
# -DSTEDM_ATTN_DIAG=<bits> (compile-time ingredient removal in attn_flash_kernel; see attn.hip). Diagnostic only; run e.g.
#   STEDM_HIP_LIB=tools/_ab/lib_attn_diag1.so python tools/bench_attn.py f16
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/_ab
objs=$(ls $R/stedm_amd/csrc/*.o | grep -v "/attn.o")
for bits in ${@:-1 2 4 8 3}; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DSTEDM_ATTN_DIAG=$bits -c $R/stedm_amd/csrc/attn.hip -o /tmp/attn_diag$bits.o &
done
wait
for bits in ${@:-1 2 4 8 3}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_ab/lib_attn_diag$bits.so $objs /tmp/attn_diag$bits.o
  echo built $R/tools/_ab/lib_attn_diag$bits.so
done
