// Boundary convolutions of the U-Net (HBM-bound, tiny K or tiny N): exact fp32 FMA.
//   conv_in : cat([x, c_concat]) NCHW  -> 3x3 conv -> NHWC   (ddpm.py:1414-1417, openaimodel.py:542)
//   conv_out: NHWC -> GN affine + SiLU -> 3x3 conv -> NCHW   (openaimodel.py:729-733, 806)
#include "common.hpp"
using namespace stedm;

// One block per output image row (b, y). LDS: 3 input rows x (W+2) x cin, and weights [tap][ci][cout].
__global__ void __launch_bounds__(256) conv_in_kernel(const float* __restrict__ x1, int c1, const float* __restrict__ x2,
                                                      int c2, int bmod, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out, int H, int W,
                                                      int cout) {
  extern __shared__ float sm[];
  const int cin = c1 + c2;
  const int PW = W + 2;
  float* sx = sm;                  // [3][PW][cin]
  float* sw = sm + 3 * PW * cin;   // [9][cin][cout]
  const int b = blockIdx.x / H, y = blockIdx.x % H;
  const int b2 = bmod > 0 ? b % bmod : b;
  for (int i = threadIdx.x; i < 3 * PW * cin; i += blockDim.x) {
    const int ci = i % cin;
    const int pc = (i / cin) % PW;
    const int dy = i / (cin * PW);
    const int sy = y + dy - 1, sxx = pc - 1;
    float v = 0.f;
    if (sy >= 0 && sy < H && sxx >= 0 && sxx < W)
      v = ci < c1 ? x1[(((long)b * c1 + ci) * H + sy) * W + sxx] : x2[(((long)b2 * c2 + (ci - c1)) * H + sy) * W + sxx];
    sx[i] = v;
  }
  for (int i = threadIdx.x; i < 9 * cin * cout; i += blockDim.x) {
    const int co = i % cout;
    const int ci = (i / cout) % cin;
    const int tap = i / (cout * cin);
    sw[i] = w[((long)co * cin + ci) * 9 + tap];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < W * cout; i += blockDim.x) {
    const int co = i % cout, x = i / cout;
    float acc = bias ? bias[co] : 0.f;
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      const float* px = sx + (dy * PW + x + dx) * cin;
      const float* pw = sw + (long)tap * cin * cout + co;
      for (int ci = 0; ci < cin; ++ci) acc = fmaf(px[ci], pw[ci * cout], acc);
    }
    out[(((long)b * H + y) * W + x) * cout + co] = acc;
  }
}

extern "C" int stedm_conv_in(const float* x1, int c1, const float* x2, int c2, int x2_bmod, const float* w,
                             const float* bias, float* out, int B, int H, int W, int cout, void* stream) {
  STEDM_CHECK_ARG(x1 && w && out, "conv_in: null pointer");
  STEDM_CHECK_ARG((x2 != nullptr) == (c2 > 0), "conv_in: x2/c2 mismatch");
  const int cin = c1 + c2;
  const size_t lds = ((size_t)3 * (W + 2) * cin + (size_t)9 * cin * cout) * sizeof(float);
  STEDM_CHECK_ARG(lds <= 64 * 1024, "conv_in: cin=%d cout=%d W=%d needs %zu B LDS (> 64 KiB)", cin, cout, W, lds);
  conv_in_kernel<<<B * H, 256, lds, as_stream(stream)>>>(x1, c1, x2, c2, x2_bmod, w, bias, out, H, W, cout);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// One block per (b, y, 32-pixel segment). Each output pixel's K = 9*c reduction is split over PARTS
// lanes (channel-interleaved: lane part takes channels part, part+PARTS, ...) and combined by shuffles.
constexpr int CO_PARTS = 8;
constexpr int CO_MAXOUT = 8;
constexpr int CO_SEG = 32;
__global__ void __launch_bounds__(256) conv_out_kernel(const float* __restrict__ src, int c, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int H, int W,
                                                       int cout, int nseg) {
  extern __shared__ float sm[];
  constexpr int PW = CO_SEG + 2;
  const int CST = c + 8;                   // padded channel stride (bank spread across pixels)
  float* sx = sm;                          // [3][PW][CST]  (post GN+SiLU)
  float* sw = sm + 3 * PW * CST;           // [cout][9][c]
  const int seg = blockIdx.x % nseg;
  const int by = blockIdx.x / nseg;
  const int b = by / H, y = by % H;
  const int x0 = seg * CO_SEG;
  const int c4 = c >> 2;
  for (int i = threadIdx.x; i < 3 * PW * c4; i += blockDim.x) {
    const int q = i % c4;
    const int pc = (i / c4) % PW;
    const int dy = i / (c4 * PW);
    const int sy = y + dy - 1, sxx = x0 + pc - 1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sy >= 0 && sy < H && sxx >= 0 && sxx < W) {
      v = *reinterpret_cast<const float4*>(src + (((long)b * H + sy) * W + sxx) * c + q * 4);
      const float4 sc = *reinterpret_cast<const float4*>(scale + (long)b * c + q * 4);
      const float4 sh = *reinterpret_cast<const float4*>(shift + (long)b * c + q * 4);
      v.x = silu_f(fmaf(v.x, sc.x, sh.x)); v.y = silu_f(fmaf(v.y, sc.y, sh.y));
      v.z = silu_f(fmaf(v.z, sc.z, sh.z)); v.w = silu_f(fmaf(v.w, sc.w, sh.w));
    }
    *reinterpret_cast<float4*>(sx + (dy * PW + pc) * CST + q * 4) = v;
  }
  for (int i = threadIdx.x; i < cout * 9 * c; i += blockDim.x) {
    const int ci = i % c;
    const int tap = (i / c) % 9;
    const int co = i / (c * 9);
    sw[i] = w[((long)co * c + ci) * 9 + tap];
  }
  __syncthreads();
  const int part = threadIdx.x % CO_PARTS;
  const int xl = threadIdx.x / CO_PARTS;   // 0..31: pixel within the segment
  const int x = x0 + xl;
  float acc[CO_MAXOUT];
#pragma unroll
  for (int o = 0; o < CO_MAXOUT; ++o) acc[o] = 0.f;
  if (x < W) {
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      const float* px = sx + (dy * PW + xl + dx) * CST;
      const float* pw = sw + tap * c;
      for (int ci = part; ci < c; ci += CO_PARTS) {
        const float v = px[ci];
#pragma unroll
        for (int o = 0; o < CO_MAXOUT; ++o)
          if (o < cout) acc[o] = fmaf(v, pw[o * 9 * c + ci], acc[o]);
      }
    }
  }
#pragma unroll
  for (int o = 0; o < CO_MAXOUT; ++o) {
    float v = acc[o];
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    if (o < cout && part == 0 && x < W) out[(((long)b * cout + o) * H + y) * W + x] = v + (bias ? bias[o] : 0.f);
  }
}

extern "C" int stedm_conv_out(const float* src, int c, const float* scale, const float* shift, const float* w,
                              const float* bias, float* out, int B, int H, int W, int cout, void* stream) {
  STEDM_CHECK_ARG(src && scale && shift && w && out, "conv_out: null pointer");
  STEDM_CHECK_ARG(c % 4 == 0 && cout >= 1 && cout <= CO_MAXOUT, "conv_out: need c %% 4 == 0 and cout <= %d (c=%d cout=%d)", CO_MAXOUT, c, cout);
  const size_t lds = ((size_t)3 * (CO_SEG + 2) * (c + 8) + (size_t)cout * 9 * c) * sizeof(float);
  STEDM_CHECK_ARG(lds <= 160 * 1024, "conv_out: needs %zu B LDS", lds);
  if (lds > 64 * 1024)
    STEDM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_out_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int nseg = (W + CO_SEG - 1) / CO_SEG;
  conv_out_kernel<<<B * H * nseg, 256, lds, as_stream(stream)>>>(src, c, scale, shift, w, bias, out, H, W, cout, nseg);
  STEDM_LAUNCH_CHECK();
  return 0;
}
