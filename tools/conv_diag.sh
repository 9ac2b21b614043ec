#!/bin/bash
# Builds tools/_ab/lib_conv_diag<bits>.so for bits in "$@" (default: 8 16 32 56): the shipped objects with the bf16 single-product convolution
# unit recompiled under -DSTEDM_CONV_DIAG=<bits> (conv_rs.inc: compile-time ingredient removal for the in-loop cycle attribution).
# Diagnostic only; the product library is untouched.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/_ab
objs=$(ls $R/stedm_amd/csrc/*.o | grep -v "/conv_dma_bf16_p1.o")
for bits in ${@:-8 16 32 56}; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DSTEDM_CONV_DIAG=$bits -c $R/stedm_amd/csrc/conv_dma_bf16_p1.hip -o /tmp/conv_dma_bf16_p1_diag$bits.o &
done
wait
for bits in ${@:-8 16 32 56}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_ab/lib_conv_diag$bits.so $objs /tmp/conv_dma_bf16_p1_diag$bits.o
  echo built $R/tools/_ab/lib_conv_diag$bits.so
done
