#!/bin/bash
# Register / scratch use of the device kernels in one object file of the build (code-object notes):
#   bash tools/kernel_resources.sh stedm_amd/csrc/conv_dma_bf16_p1.o [name filter]
set -e
O=$(realpath "$1"); F=${2:-.}
D=$(mktemp -d); cp "$O" "$D/x.o"
( cd "$D" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o > /dev/null 2>&1 || true )
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$D"/x.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 |
  awk '/^ +\.name:/ {n=$2} /^ +\.private_segment_fixed_size:/ {p=$2} /^ +\.sgpr_spill_count:/ {s=$2} /^ +\.vgpr_count:/ {v=$2} /^ +\.vgpr_spill_count:/ {print n, "vgpr", v, "vgpr_spill", $2, "sgpr_spill", s, "scratch", p}' | grep -E "$F" || true
rm -rf "$D"
