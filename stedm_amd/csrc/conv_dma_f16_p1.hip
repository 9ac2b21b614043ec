// explicit instantiation unit of the v3 DMA convolution (f16, single product); see conv_igemm_dma.inc / conv_igemm_dma9.inc
#include "conv_igemm_dma9.inc"
#include "conv_rs.inc"
namespace stedm {
int conv_dma_pick_f16_p1(ConvParams& p, hipStream_t st, bool dry) {
  int rc = conv_rs_pick<_Float16>(p, st, dry);        // fragment-order weights: weights bypass LDS
  if (rc >= 0 || dry || p.a.src16b_hi || p.a.mode == STEDM_CONV_S2D || p.a.qkv_q || p.a.ln_gamma) return rc;   // fused skip / space-to-depth / qkv and LayerNorm epilogues exist in that kernel only
  rc = dma9_pick<_Float16>(p, st);     // 3x3: one barrier per 16-channel chunk
  return rc >= 0 ? rc : dma_pick<1, _Float16>(p, st);
}
}  // namespace stedm
