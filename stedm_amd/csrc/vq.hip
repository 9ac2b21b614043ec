// First stage (VQ-f4 autoencoder) pieces that the convolution / GroupNorm kernels do not cover:
//   vq_nearest       taming VectorQuantizer2.forward as VQModelInterface.decode calls it (ldm/models/autoencoder.py:274-282): nearest codebook
//                    entry per latent pixel (integer result: index of the minimum of |z|^2 + |e|^2 - 2 z.e, first index on ties) and
//                    the straight-through output z + (e - z)
//   conv1x1_nchw     quant_conv / post_quant_conv (autoencoder.py:42-43): 1x1 convolutions over a handful of channels, NCHW
//   softmax_rows16   the softmax of AttnBlock (ldm/modules/diffusionmodules/model.py:143-199: single head of width C, logits scaled by
//                    C^-0.5) written as 16-bit operand planes for the P @ V GEMM
//   nearest-codebook arithmetic is written WITHOUT fused multiply-adds and in a fixed order (the index is an integer
//   result and must not depend on contraction choices of the compiler)
#include "common.hpp"
using namespace stedm;

namespace {

constexpr int VQ_MAXE = 8;       // embedding width (3 in conf/diffusion/first_stage_config/vq-f4.yaml; 4 for the 4-channel synthetic latents)

// codebook [n_e][e] fp32, z NCHW [B][e][HW]. One thread per latent pixel; the codebook walks through LDS in tiles of 1024 entries
// (+ the squared norm).
template <int E>
__global__ void __launch_bounds__(256) vq_nearest_kernel(const float* __restrict__ z, const float* __restrict__ cb, int n_e, long npix, long HW,
                                                         long long* __restrict__ idx_out, float* __restrict__ zq) {
#pragma clang fp contract(off)
  constexpr int TILE = 1024;
  __shared__ float scb[TILE][E + 1];
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = p < npix;
  const long b = live ? p / HW : 0, hw = live ? p - b * HW : 0;
  float zv[E];
  float zz = 0.f;
#pragma unroll
  for (int c = 0; c < E; ++c) {
    zv[c] = live ? z[(b * E + c) * HW + hw] : 0.f;
    const float sq = zv[c] * zv[c];
    zz = c == 0 ? sq : zz + sq;                      // torch.sum(z**2, dim=1): left-to-right fp32 sum
  }
  float best = 0.f;
  int bi = -1;
  for (int t0 = 0; t0 < n_e; t0 += TILE) {
    const int nt = min(TILE, n_e - t0);
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += 256) {
      float ee = 0.f;
#pragma unroll
      for (int c = 0; c < E; ++c) {
        const float v = cb[(long)(t0 + i) * E + c];
        scb[i][c] = v;
        const float sq = v * v;
        ee = c == 0 ? sq : ee + sq;
      }
      scb[i][E] = ee;
    }
    __syncthreads();
    if (live) {
      for (int i = 0; i < nt; ++i) {
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < E; ++c) {
          const float pr = zv[c] * scb[i][c];
          dot = c == 0 ? pr : dot + pr;
        }
        const float d = (zz + scb[i][E]) - 2.0f * dot;
        if (bi < 0 || d < best) { best = d; bi = t0 + i; }   // strict <: the first index wins a tie (torch.argmin)
      }
    }
  }
  if (!live) return;
  idx_out[p] = bi;
#pragma unroll
  for (int c = 0; c < E; ++c) {
    const float e = cb[(long)bi * E + c];
    zq[(b * E + c) * HW + hw] = zv[c] + (e - zv[c]);    // z + (z_q - z).detach(): the straight-through form's forward value
  }
}
// out[b][co][p] = bias[co] + sum_ci w[co][ci] * x[b][ci][p]; cin, cout <= 16
__global__ void __launch_bounds__(256) conv1x1_nchw_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ out, int cin, int cout, long HW, long npix) {
  __shared__ float sw[16 * 16 + 16];
  for (int i = threadIdx.x; i < cin * cout; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < cout; i += 256) sw[256 + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  const long b = p / HW, hw = p - b * HW;
  float xv[16];
  for (int c = 0; c < cin; ++c) xv[c] = x[(b * cin + c) * HW + hw];
  for (int co = 0; co < cout; ++co) {
    float acc = sw[256 + co];
    for (int c = 0; c < cin; ++c) acc = fmaf(sw[co * cin + c], xv[c], acc);
    out[(b * cout + co) * HW + hw] = acc;
  }
}

// one wave per row: softmax(scale * x[row][:n]) -> 16-bit hi (/lo) planes with row stride ld_out (columns n..ld_out-1 are zeroed: the
// GEMM that follows contracts over whole 64-element chunks)
template <typename T>
__global__ void __launch_bounds__(256) softmax_rows16_kernel(const float* __restrict__ x, long ld_in, float scale, T* __restrict__ hi, T* __restrict__ lo,
                                                             long rows, int n, long ld_out) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* px = x + row * ld_in;
  float m = -3.0e38f;
  for (int k = lane; k < n; k += 64) m = fmaxf(m, px[k] * scale);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float s = 0.f;
  for (int k = lane; k < n; k += 64) s += __expf(px[k] * scale - m);
  const float inv = 1.0f / wave_sum(s);
  for (int k = lane; k < ld_out; k += 64) {
    const float v = k < n ? __expf(px[k] * scale - m) * inv : 0.f;
    const T h = (T)v;
    hi[row * ld_out + k] = h;
    if (lo) lo[row * ld_out + k] = (T)(v - (float)h);
  }
}

}  // namespace

extern "C" int stedm_vq_nearest(const float* z, const float* codebook, int n_e, int e_dim, int B, long HW, long long* idx, float* zq, void* stream) {
  STEDM_CHECK_ARG(z && codebook && idx && zq && n_e > 0 && B > 0 && HW > 0, "vq_nearest: bad args");
  STEDM_CHECK_ARG(e_dim >= 1 && e_dim <= VQ_MAXE, "vq_nearest: embedding width %d unsupported (1..%d)", e_dim, VQ_MAXE);
  const long npix = (long)B * HW;
  const int grid = (int)((npix + 255) / 256);
  hipStream_t st = as_stream(stream);
  switch (e_dim) {
#define CASE(E) case E: vq_nearest_kernel<E><<<grid, 256, 0, st>>>(z, codebook, n_e, npix, HW, idx, zq); break;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
  }
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_conv1x1_nchw(const float* x, const float* w, const float* bias, float* out, int B, int cin, int cout, long HW, void* stream) {
  STEDM_CHECK_ARG(x && w && out && B > 0 && HW > 0 && cin >= 1 && cin <= 16 && cout >= 1 && cout <= 16, "conv1x1_nchw: bad args (cin, cout <= 16)");
  const long npix = (long)B * HW;
  conv1x1_nchw_kernel<<<(int)((npix + 255) / 256), 256, 0, as_stream(stream)>>>(x, w, bias, out, cin, cout, HW, npix);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_softmax_rows16(const float* x, long ld_in, float scale, void* out_hi, void* out_lo, long rows, int n, long ld_out, int mm_dtype,
                                    void* stream) {
  STEDM_CHECK_ARG(x && out_hi && rows > 0 && n > 0 && ld_in >= n && ld_out >= n, "softmax_rows16: bad args");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "softmax_rows16: bad mm_dtype %d", mm_dtype);
  const int grid = (int)((rows + 3) / 4);
  if (mm_dtype == STEDM_F16)
    softmax_rows16_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>(x, ld_in, scale, (_Float16*)out_hi, (_Float16*)out_lo, rows, n, ld_out);
  else
    softmax_rows16_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>(x, ld_in, scale, (__bf16*)out_hi, (__bf16*)out_lo, rows, n, ld_out);
  STEDM_LAUNCH_CHECK();
  return 0;
}
