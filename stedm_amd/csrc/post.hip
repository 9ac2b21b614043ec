// Epilogue of LDM_Diffusion.predict_step (modules/ldm_diffusion.py:93-99): integer work, bit-exact against numpy.
#include "common.hpp"

using namespace stedm;

namespace {

// out[b][y][x][c] = uint8(trunc((clip(x[b][c][y][x], -1, 1) + 1) * 127.5)) with float32 arithmetic in numpy's order (add, then multiply:
// no FMA contraction, or the rounding of the sum is lost)
__global__ void image_to_uint8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int C, int HW, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over the NHWC output
  if (i >= total) return;
  const int c = (int)(i % C);
  const long p = (i / C) % HW, b = i / ((long)C * HW);
  float v = x[(b * C + c) * HW + p];
  v = fminf(fmaxf(v, -1.0f), 1.0f);
  const float s = __fmul_rn(__fadd_rn(v, 1.0f), 127.5f);
  out[i] = (uint8_t)(int)s;      // C-style truncation, as ndarray.astype(np.uint8) on values in [0, 255]
}

// seg [N][ncls] fp32 (NHWC one-hot / logits) -> uint8 index of the first maximum (torch.argmax(dim=-1))
__global__ void argmax_u8_kernel(const float* __restrict__ seg, uint8_t* __restrict__ out, int ncls, long N) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float* p = seg + i * ncls;
  int best = 0;
  float bv = p[0];
  for (int k = 1; k < ncls; ++k)
    if (p[k] > bv) { bv = p[k]; best = k; }
  out[i] = (uint8_t)best;
}

// LDM_Diffusion.prepare_batch's segmentation handling (modules/ldm_diffusion.py:52-56): seg NCHW [B][K][H][W] one-hot ->
// NHWC [B][H][W][2] with channel 0 = class 0 and channel 1 = sum of classes 1..K-1 (in class order, like torch.sum over the last dim)
__global__ void seg_merge_kernel(const float* __restrict__ seg, float* __restrict__ out, int K, long HW, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over B * HW pixels
  if (i >= total) return;
  const long b = i / HW, p = i - b * HW;
  const float* s = seg + b * K * HW + p;
  float fg = 0.f;
  for (int k = 1; k < K; ++k) fg += s[(long)k * HW];
  out[i * 2] = s[0];
  out[i * 2 + 1] = fg;
}

}  // namespace

extern "C" int stedm_seg_merge(const float* seg, float* out, int B, int K, int H, int W, void* stream) {
  STEDM_CHECK_ARG(seg && out && B > 0 && K >= 2 && H > 0 && W > 0, "seg_merge: bad args (K >= 2)");
  const long total = (long)B * H * W;
  seg_merge_kernel<<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(seg, out, K, (long)H * W, total);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_image_to_uint8(const float* x, unsigned char* out, int B, int C, int H, int W, void* stream) {
  STEDM_CHECK_ARG(x && out && B > 0 && C > 0 && H > 0 && W > 0, "image_to_uint8: bad args");
  const long total = (long)B * C * H * W;
  image_to_uint8_kernel<<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(x, out, C, H * W, total);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_argmax_u8(const float* seg, unsigned char* out, long N, int ncls, void* stream) {
  STEDM_CHECK_ARG(seg && out && N > 0 && ncls > 0 && ncls <= 256, "argmax_u8: bad args");
  argmax_u8_kernel<<<(unsigned)((N + 255) / 256), 256, 0, as_stream(stream)>>>(seg, out, ncls, N);
  STEDM_LAUNCH_CHECK();
  return 0;
}
