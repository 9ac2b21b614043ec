// explicit instantiation unit of the v3 DMA convolution (f16, 1 product); see conv_igemm_dma.inc
#include "conv_igemm_dma.inc"
namespace stedm { int conv_dma_pick_f16_p1(ConvParams& p, hipStream_t st) { return dma_pick<1, _Float16>(p, st); } }
