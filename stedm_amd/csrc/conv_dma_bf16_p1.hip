// explicit instantiation unit of the v3 DMA convolution (bf16, single product); see conv_igemm_dma.inc / conv_igemm_dma9.inc
#include "conv_igemm_dma9.inc"
#include "conv_rs.inc"
namespace stedm {
int conv_dma_pick_bf16_p1(ConvParams& p, hipStream_t st, bool dry) {
  int rc = conv_rs_pick<__bf16>(p, st, dry);        // fragment-order weights: weights bypass LDS
  if (rc >= 0 || dry || p.a.src16b_hi || p.a.mode == STEDM_CONV_S2D || p.a.qkv_q || p.a.ln_gamma) return rc;   // fused skip / space-to-depth / qkv and LayerNorm epilogues exist in that kernel only
  rc = dma9_pick<__bf16>(p, st);     // 3x3: one barrier per 16-channel chunk
  return rc >= 0 ? rc : dma_pick<1, __bf16>(p, st);
}
}  // namespace stedm

// diagnostics (timing experiments, STEDM_CONV_DBG & 1024): phase stamps of the last bf16 conv_rs_kernel launch, 8 per block
extern "C" int stedm_debug_conv_stamps(unsigned long long* host_out, int nblocks) {
  if (!host_out || nblocks <= 0 || nblocks > 2048) { stedm::set_error("debug_conv_stamps: bad args"); return 1; }
  hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_conv_stamps), (size_t)nblocks * 8 * sizeof(unsigned long long));
  if (e != hipSuccess) { stedm::set_error("debug_conv_stamps: %s", hipGetErrorString(e)); return 2; }
  return 0;
}
