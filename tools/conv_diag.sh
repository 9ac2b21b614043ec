#!/bin/bash
# Builds tools/_ab/lib_conv_diag<bits>.so for bits in "$@" (default: 8 16 32 56): the shipped objects with the convolution dispatcher, the bf16
# and f16 single-product convolution units and the weight-gradient unit recompiled under -DSTEDM_CONV_DIAG=<bits>:
#   * bits 8 / 16 / 32 (conv_rs.inc RS_DIAG): compile-time ingredient removal for the in-loop cycle attribution;
#   * any value, 0 included: the run-time switches STEDM_CONV_DBG / STEDM_WGRAD_DBG and the phase stamps (conv_common.hpp STEDM_DBG) that the
#     shipped library compiles out (tools/conv_phases.py, tools/gemm_phases.py, tools/bench_conv_abl.py: run them with STEDM_HIP_LIB=<this>).
# Diagnostic only; the product library is untouched.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/_ab
units="conv_igemm conv_dma_bf16_p1 conv_dma_f16_p1 wgrad"
objs=$(ls $R/stedm_amd/csrc/*.o | grep -v -e "/conv_igemm.o" -e "/conv_dma_bf16_p1.o" -e "/conv_dma_f16_p1.o" -e "/wgrad.o")
for bits in ${@:-8 16 32 56}; do
  for u in $units; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DSTEDM_CONV_DIAG=$bits -c $R/stedm_amd/csrc/$u.hip -o /tmp/${u}_diag$bits.o &
  done
done
wait
for bits in ${@:-8 16 32 56}; do
  dobjs=""
  for u in $units; do dobjs="$dobjs /tmp/${u}_diag$bits.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_ab/lib_conv_diag$bits.so $objs $dobjs
  echo built $R/tools/_ab/lib_conv_diag$bits.so
done
