// explicit instantiation unit of the v3 DMA convolution (bf16, 3 products); see conv_igemm_dma.inc / conv_rs.inc
#include "conv_rs.inc"
namespace stedm {
int conv_dma_pick_bf16_p3(ConvParams& p, hipStream_t st, bool dry) {
  const int rc = conv_rs3_pick<__bf16>(p, st, dry);      // hi + lo fragment streams given: register-streamed weights, 16x16x32 MFMA
  return (rc >= 0 || dry || p.a.mode == STEDM_CONV_S2D) ? rc : dma_pick<3, __bf16>(p, st);     // (the space-to-depth form exists in that kernel only)
}
}  // namespace stedm
