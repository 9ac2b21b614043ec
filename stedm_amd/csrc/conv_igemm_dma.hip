// v3 DMA convolution: dispatcher. The kernel template lives in conv_igemm_dma.inc and is instantiated in
// conv_dma_{f16,bf16}_p{1,3}.hip (separate translation units so that hipcc compiles them in parallel).
#include "conv_common.hpp"
using namespace stedm;

namespace stedm {
int conv_dma_pick_f16_p1(ConvParams&, hipStream_t, bool dry);
int conv_dma_pick_f16_p3(ConvParams&, hipStream_t, bool dry);
int conv_dma_pick_bf16_p1(ConvParams&, hipStream_t, bool dry);
int conv_dma_pick_bf16_p3(ConvParams&, hipStream_t, bool dry);
}  // namespace stedm

int stedm::conv_launch_dma(ConvParams& p, hipStream_t st, bool dry) {
  const stedm_conv_args& a = p.a;
  if (a.src16b_hi && a.npass != 1) { set_error("conv_igemm(dma): the fused skip phase is single-product only"); return 1; }
  if (dry && a.npass != 1 && !a.w_frag16) return 1;
  if (a.src16b_hi && ((long)a.B * a.Hin * a.Win * a.cb >= (1L << 31) || a.cb * 2 + 256 > STEDM_ZERO_PAGE_BYTES)) {
    set_error("conv_igemm(dma): fused skip operand too large (cb=%d)", a.cb);
    return 1;
  }
  if ((long)a.B * a.Hin * a.Win * p.Cin >= (1L << 31)) {
    set_error("conv_igemm(dma): activation tensor too large for 32-bit element offsets");
    return 1;
  }
  if (p.Cin * 2 + 256 > STEDM_ZERO_PAGE_BYTES) { set_error("conv_igemm(dma): Cin too large for the zero page"); return 1; }
  const bool f16 = a.mm_dtype == STEDM_F16;
  int rc = a.npass == 3 ? (f16 ? conv_dma_pick_f16_p3(p, st, dry) : conv_dma_pick_bf16_p3(p, st, dry))
                        : (f16 ? conv_dma_pick_f16_p1(p, st, dry) : conv_dma_pick_bf16_p1(p, st, dry));
  if (dry) return rc < 0 ? 1 : 0;
  if (rc < 0 && a.ln_gamma) { set_error("conv_igemm(dma): the register-streamed kernel does not run this LayerNorm-epilogue problem (see stedm_conv_rs_ok)"); return 1; }
  if (rc < 0 && a.qkv_q) { set_error("conv_igemm(dma): the register-streamed kernel does not run this qkv-epilogue problem (see stedm_conv_rs_ok)"); return 1; }
  if (rc < 0 && a.src16b_hi) { set_error("conv_igemm(dma): no kernel runs this fused skip problem (see stedm_conv_fused_skip_ok)"); return 1; }
  if (rc < 0) { set_error("conv_igemm(dma): no tile configuration fits (Hin=%d Win=%d mode=%d Cin=%d)", a.Hin, a.Win, a.mode, p.Cin); return 1; }
  return rc;
}
