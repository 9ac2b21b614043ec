// explicit instantiation unit of the v3 DMA convolution (f16, 3 products); see conv_igemm_dma.inc
#include "conv_igemm_dma.inc"
namespace stedm { int conv_dma_pick_f16_p3(ConvParams& p, hipStream_t st) { return dma_pick<3, _Float16>(p, st); } }
