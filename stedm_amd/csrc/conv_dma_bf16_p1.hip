// explicit instantiation unit of the v3 DMA convolution (bf16, 1 product); see conv_igemm_dma.inc
#include "conv_igemm_dma.inc"
namespace stedm { int conv_dma_pick_bf16_p1(ConvParams& p, hipStream_t st) { return dma_pick<1, __bf16>(p, st); } }
