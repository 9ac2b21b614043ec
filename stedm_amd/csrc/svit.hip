// Set-ViT style encoder (networks/vit_set.py) and the small style-path kernels (agg blocks, SpatialRescaler).
//
//   svit_patch_embed : SPT (vit_set.py:84-107): channel-stack the set, 8x8 patches '(p1 p2 c)', LayerNorm, Linear,
//                      + pos_embedding, written at token index 2.. (tokens 0/1 = cls / zero time token, :175-186)
//   ln_apply16       : PreNorm LayerNorm (vit_set.py:14-20) -> 16-bit operand planes for the MFMA GEMMs
//   qkv_pack         : to_qkv output chunks [q|k|v] '(h d)' (vit_set.py:53-54) -> per-head q/k rows and V^T, logits
//                      scale exp(temperature) folded into q (vit_set.py:56)
//   lsa_flash        : softmax(mask_diag(q k^T)) v (vit_set.py:56-66) flash-style on MFMA, never materialising T x T
//   svit_head        : pool (mean / sum / cls) + mlp_head LayerNorm + Linear (vit_set.py:191-206)
//   agg_reduce       : Agg_Mean / Agg_Max over the set dimension (agg_blocks.py:52,73)
//   spatial_rescale  : SpatialRescaler (encoders/modules.py:123-130): n x bilinear 1/2 (== 2x2 box mean for even sizes)
//                      then bias-free 1x1 conv
#include <float.h>
#include <stdlib.h>

#include "conv_common.hpp"
#include "dropmask.hpp"
using namespace stedm;

#define GLDS16(gptr, lptr)                                                                                  \
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr),                   \
                                   (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

// ------------------------------------------------------------------------------------------------ patch embed
struct PatchArgs {
  const float* img;   // [B][ns][H][W][3]
  const float* ln_w;  // [pd]
  const float* ln_b;
  const float* wt;    // [pd][dim]
  const float* bias;  // [dim]
  const float* pos;   // [ntok+2][dim]
  const float* cls;   // [dim]
  float* x;           // [B][ntok+2][dim]
  int ns, H, W, p, dim, tg;
  float eps;
};

__global__ void __launch_bounds__(256) svit_patch_embed_kernel(PatchArgs a) {
  extern __shared__ float sf[];   // [tg][pd]
  const int pw = a.W / a.p, ph = a.H / a.p;
  const int C = 3 * a.ns, pd = a.p * a.p * C;
  const int groups_per_row = pw / a.tg;
  const int b = blockIdx.x / (ph * groups_per_row);
  const int rem = blockIdx.x % (ph * groups_per_row);
  const int hp = rem / groups_per_row, w0 = (rem % groups_per_row) * a.tg;
  const int ntok = ph * pw;
  // gather: for every image s and patch row p1 one contiguous run of tg*p pixels x 3 channels
  const int run = a.tg * a.p * 3;
  for (int i = threadIdx.x; i < a.ns * a.p * run; i += 256) {
    const int e = i % run;
    const int p1 = (i / run) % a.p;
    const int s = i / (run * a.p);
    const int px = e / 3, c = e - px * 3;
    const int t = px / a.p, p2 = px - t * a.p;
    const float v = a.img[((((long)b * a.ns + s) * a.H + hp * a.p + p1) * a.W + w0 * a.p) * 3 + e];
    sf[t * pd + (p1 * a.p + p2) * C + c * a.ns + s] = v;   // stacked channel = c*ns + s (vit_set.py:105-106)
  }
  __syncthreads();
  // LayerNorm per token (wave w handles tokens w, w+4, ...)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int t = wave; t < a.tg; t += 4) {
    float s1 = 0.f;
    for (int k = lane; k < pd; k += 64) s1 += sf[t * pd + k];
    const float mean = wave_sum(s1) / pd;
    float s2 = 0.f;
    for (int k = lane; k < pd; k += 64) { const float d = sf[t * pd + k] - mean; s2 += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / pd + a.eps);
    for (int k = lane; k < pd; k += 64) sf[t * pd + k] = (sf[t * pd + k] - mean) * rstd * a.ln_w[k] + a.ln_b[k];
  }
  __syncthreads();
  for (int n = threadIdx.x; n < a.dim; n += 256) {
    float acc[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) acc[t] = 0.f;
    for (int k = 0; k < pd; ++k) {
      const float w = a.wt[(long)k * a.dim + n];
#pragma unroll
      for (int t = 0; t < 32; ++t)
        if (t < a.tg) acc[t] = fmaf(sf[t * pd + k], w, acc[t]);
    }
    const float bv = a.bias[n];
#pragma unroll
    for (int t = 0; t < 32; ++t)
      if (t < a.tg) {
        const int tok = hp * pw + w0 + t + 2;
        a.x[((long)b * (ntok + 2) + tok) * a.dim + n] = acc[t] + bv + a.pos[(long)tok * a.dim + n];
      }
    if (rem == 0) {   // cls token and the zero time token (t_emb=None, vit_set.py:177-178)
      a.x[((long)b * (ntok + 2) + 0) * a.dim + n] = a.cls[n] + a.pos[n];
      a.x[((long)b * (ntok + 2) + 1) * a.dim + n] = a.pos[a.dim + n];
    }
  }
}

extern "C" int stedm_svit_patch_embed(const float* img, int B, int ns, int H, int W, int patch, const float* ln_w,
                                      const float* ln_b, float eps, const float* wt, const float* bias, const float* pos,
                                      const float* cls, float* x, int dim, void* stream) {
  STEDM_CHECK_ARG(img && ln_w && ln_b && wt && bias && pos && cls && x, "svit_patch_embed: null pointer");
  STEDM_CHECK_ARG(H % patch == 0 && W % patch == 0, "svit_patch_embed: image not divisible by patch");
  const int pd = patch * patch * 3 * ns, pw = W / patch;
  int tg = 32;
  while (tg > 1 && ((size_t)tg * pd * 4 > 96 * 1024 || pw % tg != 0)) tg >>= 1;
  STEDM_CHECK_ARG(pw % tg == 0 && (size_t)tg * pd * 4 <= 160 * 1024, "svit_patch_embed: patch_dim %d too large", pd);
  PatchArgs a{img, ln_w, ln_b, wt, bias, pos, cls, x, ns, H, W, patch, dim, tg, eps};
  const size_t lds = (size_t)tg * pd * sizeof(float);
  if (lds > 64 * 1024)
    STEDM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(svit_patch_embed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  svit_patch_embed_kernel<<<B * (H / patch) * (pw / tg), 256, lds, as_stream(stream)>>>(a);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// MFMA form of the patch embedding: (1) gather + LayerNorm of the patch features -> 16-bit operand planes [B*ntok][pd],
// (2) the Linear as a 1x1 GEMM (stedm_conv_igemm), (3) + pos_embedding, placed at token 2.., cls / zero time token written.
template <typename T>
__global__ void __launch_bounds__(256) svit_patch_ln16_kernel(PatchArgs a, T* __restrict__ hi, T* __restrict__ lo) {
  extern __shared__ float sf[];   // [tg][pd]
  const int pw = a.W / a.p, ph = a.H / a.p;
  const int C = 3 * a.ns, pd = a.p * a.p * C;
  const int groups_per_row = pw / a.tg;
  const int b = blockIdx.x / (ph * groups_per_row);
  const int rem = blockIdx.x % (ph * groups_per_row);
  const int hp = rem / groups_per_row, w0 = (rem % groups_per_row) * a.tg;
  const int ntok = ph * pw;
  const int run = a.tg * a.p * 3;
  for (int i = threadIdx.x; i < a.ns * a.p * run; i += 256) {
    const int e = i % run;
    const int p1 = (i / run) % a.p;
    const int s = i / (run * a.p);
    const int px = e / 3, c = e - px * 3;
    const int t = px / a.p, p2 = px - t * a.p;
    const float v = a.img[((((long)b * a.ns + s) * a.H + hp * a.p + p1) * a.W + w0 * a.p) * 3 + e];
    sf[t * pd + (p1 * a.p + p2) * C + c * a.ns + s] = v;   // stacked channel = c*ns + s (vit_set.py:105-106)
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int t = wave; t < a.tg; t += 4) {
    float s1 = 0.f;
    for (int k = lane; k < pd; k += 64) s1 += sf[t * pd + k];
    const float mean = wave_sum(s1) / pd;
    float s2 = 0.f;
    for (int k = lane; k < pd; k += 64) { const float d = sf[t * pd + k] - mean; s2 += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / pd + a.eps);
    const long row = (long)b * ntok + hp * pw + w0 + t;
    for (int k = lane; k < pd; k += 64) {
      const float v = (sf[t * pd + k] - mean) * rstd * a.ln_w[k] + a.ln_b[k];
      const T h = (T)v;
      hi[row * pd + k] = h;
      if (lo) lo[row * pd + k] = (T)(v - (float)h);
    }
  }
}

extern "C" int stedm_svit_patch_ln16(const float* img, int B, int ns, int H, int W, int patch, const float* ln_w, const float* ln_b,
                                     float eps, void* out_hi, void* out_lo, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(img && ln_w && ln_b && out_hi, "svit_patch_ln16: null pointer");
  STEDM_CHECK_ARG(H % patch == 0 && W % patch == 0, "svit_patch_ln16: image not divisible by patch");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "svit_patch_ln16: bad mm_dtype");
  const int pd = patch * patch * 3 * ns, pw = W / patch;
  int tg = 8;     // tokens per block: keeps several blocks per CU
  while (tg > 1 && ((size_t)tg * pd * 4 > 48 * 1024 || pw % tg != 0)) tg >>= 1;
  STEDM_CHECK_ARG(pw % tg == 0 && (size_t)tg * pd * 4 <= 64 * 1024, "svit_patch_ln16: patch_dim %d too large", pd);
  PatchArgs a{img, ln_w, ln_b, nullptr, nullptr, nullptr, nullptr, nullptr, ns, H, W, patch, 0, tg, eps};
  const size_t lds = (size_t)tg * pd * sizeof(float);
  const int grid = B * (H / patch) * (pw / tg);
  if (mm_dtype == STEDM_F16) svit_patch_ln16_kernel<_Float16><<<grid, 256, lds, as_stream(stream)>>>(a, (_Float16*)out_hi, (_Float16*)out_lo);
  else svit_patch_ln16_kernel<__bf16><<<grid, 256, lds, as_stream(stream)>>>(a, (__bf16*)out_hi, (__bf16*)out_lo);
  STEDM_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(256) svit_tok_place_kernel(const float* __restrict__ tok, const float* __restrict__ pos,
                                                             const float* __restrict__ cls, float* __restrict__ x, int ntok, int dim4, long total4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const int n4 = (int)(i % dim4);
    const long row = i / dim4;                  // over B * (ntok + 2)
    const int t = (int)(row % (ntok + 2));
    const long b = row / (ntok + 2);
    const float4 p = reinterpret_cast<const float4*>(pos)[(long)t * dim4 + n4];
    float4 v;
    if (t >= 2) v = reinterpret_cast<const float4*>(tok)[(b * ntok + t - 2) * dim4 + n4];
    else if (t == 0) v = reinterpret_cast<const float4*>(cls)[n4];
    else v = make_float4(0.f, 0.f, 0.f, 0.f);    // zero time token (t_emb = None, vit_set.py:177-178)
    reinterpret_cast<float4*>(x)[i] = make_float4(v.x + p.x, v.y + p.y, v.z + p.z, v.w + p.w);
  }
}

extern "C" int stedm_svit_tok_place(const float* tok, const float* pos, const float* cls, float* x, int B, int ntok, int dim, void* stream) {
  STEDM_CHECK_ARG(tok && pos && cls && x && B > 0 && ntok > 0 && dim > 0 && dim % 4 == 0, "svit_tok_place: bad args (dim %% 4)");
  const long total4 = (long)B * (ntok + 2) * (dim / 4);
  const int grid = (int)((total4 + 255) / 256 < 16384 ? (total4 + 255) / 256 : 16384);
  svit_tok_place_kernel<<<grid, 256, 0, as_stream(stream)>>>(tok, pos, cls, x, ntok, dim / 4, total4);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ LayerNorm -> 16-bit
template <typename T>
__global__ void __launch_bounds__(256) ln_apply16_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                         const float* __restrict__ bt, float eps, T* __restrict__ hi,
                                                         T* __restrict__ lo, long rows, int dim) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* px = x + row * dim;
  float s1 = 0.f;
  for (int k = lane; k < dim; k += 64) s1 += px[k];
  const float mean = wave_sum(s1) / dim;
  float s2 = 0.f;
  for (int k = lane; k < dim; k += 64) { const float d = px[k] - mean; s2 += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(s2) / dim + eps);
  for (int k = lane; k < dim; k += 64) {
    const float v = (px[k] - mean) * rstd * g[k] + bt[k];
    const T h = (T)v;
    hi[row * dim + k] = h;
    if (lo) lo[row * dim + k] = (T)(v - (float)h);
  }
}

// dim = 256 NJ: a wave owns R whole rows in registers (a lane holds 4 consecutive features of every 256: one 16-B load and one 8-B store per
// plane each), one pass over HBM: R * NJ 16-B loads per lane are in flight before the first reduction (the scalar form above walks the row
// three times with 4-B accesses: 2.0 TB/s on the set-ViT's [262 k][256] residual stream). Same formulas: mean, then the centred variance.
template <typename T, int NJ, int R>
__global__ void __launch_bounds__(256) ln_apply16_vec_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                             const float* __restrict__ bt, float eps, T* __restrict__ hi,
                                                             T* __restrict__ lo, long rows) {
  typedef T V4T __attribute__((ext_vector_type(4)));
  constexpr int dim = 256 * NJ;
  const int lane = threadIdx.x & 63;
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  float4 v[R][NJ];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long row = row0 + r < rows ? row0 + r : rows - 1;      // the tail re-reads the last row (values unused)
#pragma unroll
    for (int j = 0; j < NJ; ++j) v[r][j] = *reinterpret_cast<const float4*>(x + row * dim + j * 256 + lane * 4);
  }
  float4 gm[NJ], bb[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    gm[j] = *reinterpret_cast<const float4*>(g + j * 256 + lane * 4);
    bb[j] = *reinterpret_cast<const float4*>(bt + j * 256 + lane * 4);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) s1 += (v[r][j].x + v[r][j].y) + (v[r][j].z + v[r][j].w);
    const float mean = wave_sum(s1) / dim;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float a0 = v[r][j].x - mean, a1 = v[r][j].y - mean, a2 = v[r][j].z - mean, a3 = v[r][j].w - mean;
      s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / dim + eps);
    if (row0 + r >= rows) break;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float w0 = (v[r][j].x - mean) * rstd * gm[j].x + bb[j].x, w1 = (v[r][j].y - mean) * rstd * gm[j].y + bb[j].y;
      const float w2 = (v[r][j].z - mean) * rstd * gm[j].z + bb[j].z, w3 = (v[r][j].w - mean) * rstd * gm[j].w + bb[j].w;
      V4T h4; h4[0] = (T)w0; h4[1] = (T)w1; h4[2] = (T)w2; h4[3] = (T)w3;
      const long o = (row0 + r) * dim + j * 256 + lane * 4;
      *reinterpret_cast<V4T*>(hi + o) = h4;
      if (lo) {
        V4T l4; l4[0] = (T)(w0 - (float)h4[0]); l4[1] = (T)(w1 - (float)h4[1]); l4[2] = (T)(w2 - (float)h4[2]); l4[3] = (T)(w3 - (float)h4[3]);
        *reinterpret_cast<V4T*>(lo + o) = l4;
      }
    }
  }
}

template <typename T>
static bool ln_apply16_vec(const float* x, const float* g, const float* bt, float eps, void* hi, void* lo, long rows, int dim, hipStream_t st) {
  if (dim % 256 != 0 || dim > 2048) return false;
#define STEDM_LN_VEC(NJ, R)                                                                                                          \
  ln_apply16_vec_kernel<T, NJ, R><<<(int)((rows + 4 * (R) - 1) / (4 * (R))), 256, 0, st>>>(x, g, bt, eps, (T*)hi, (T*)lo, rows)
  switch (dim / 256) {
    case 1: STEDM_LN_VEC(1, 4); break;
    case 2: STEDM_LN_VEC(2, 2); break;
    case 3: STEDM_LN_VEC(3, 1); break;
    case 4: STEDM_LN_VEC(4, 1); break;
    case 5: STEDM_LN_VEC(5, 1); break;
    case 6: STEDM_LN_VEC(6, 1); break;
    case 7: STEDM_LN_VEC(7, 1); break;
    default: STEDM_LN_VEC(8, 1); break;
  }
#undef STEDM_LN_VEC
  return true;
}

extern "C" int stedm_ln_apply16(const float* x, const float* gamma, const float* beta, float eps, void* out_hi, void* out_lo,
                                long rows, int dim, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(x && gamma && beta && out_hi && rows > 0 && dim > 0, "ln_apply16: bad args");
  if (mm_dtype == STEDM_F16 ? ln_apply16_vec<_Float16>(x, gamma, beta, eps, out_hi, out_lo, rows, dim, as_stream(stream))
                            : ln_apply16_vec<__bf16>(x, gamma, beta, eps, out_hi, out_lo, rows, dim, as_stream(stream))) {
    STEDM_LAUNCH_CHECK();
    return 0;
  }
  const int grid = (int)((rows + 3) / 4);
  if (mm_dtype == STEDM_F16)
    ln_apply16_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>(x, gamma, beta, eps, (_Float16*)out_hi, (_Float16*)out_lo, rows, dim);
  else
    ln_apply16_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>(x, gamma, beta, eps, (__bf16*)out_hi, (__bf16*)out_lo, rows, dim);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ qkv pack
// qkv fp32 [B][T][3*H*64] -> q16/k16 [B*H][Tp][64] (rows >= T zero), vT16 [B*H][64][Tp] (cols >= T zero)
template <typename T>
__global__ void __launch_bounds__(256) qkv_pack_kernel(const float* __restrict__ qkv, float qscale, T* __restrict__ qh,
                                                       T* __restrict__ ql, T* __restrict__ kh, T* __restrict__ kl,
                                                       T* __restrict__ vh, T* __restrict__ vl, int Tn, int Tp, int H) {
  __shared__ float sv[64][65];
  const int bh = blockIdx.x, b = bh / H, hd = bh % H;
  const int t0 = blockIdx.y * 64;
  const int HD = H * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int tl = i >> 6, d = i & 63;
    const int t = t0 + tl;
    float q = 0.f, k = 0.f, v = 0.f;
    if (t < Tn) {
      const float* p = qkv + ((long)b * Tn + t) * (3 * HD) + hd * 64 + d;
      q = p[0] * qscale; k = p[HD]; v = p[2 * HD];
    }
    const long o = ((long)bh * Tp + t) * 64 + d;
    const T q16 = (T)q, k16 = (T)k;
    qh[o] = q16; kh[o] = k16;
    if (ql) { ql[o] = (T)(q - (float)q16); kl[o] = (T)(k - (float)k16); }
    sv[tl][d] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int d = i >> 6, tl = i & 63;
    const float v = sv[tl][d];
    const long o = ((long)bh * 64 + d) * Tp + t0 + tl;
    const T v16 = (T)v;
    vh[o] = v16;
    if (vl) vl[o] = (T)(v - (float)v16);
  }
}

// qkv arrives as the 16-bit plane the to_qkv GEMM's epilogue wrote (single-product modes: same operand rounding for k and v, one more
// rounding of the scaled q; half the bytes of the encoder's largest activation both ways), with 16-B accesses on both sides (the scalar form above moves 2 B per lane and store): block = (sample-head, 64
// tokens); q, k: 4 threads per token, 16 channels each; V^T through an LDS tile of the MFMA type, 8 tokens per 16-B store.
template <typename T>
__global__ void __launch_bounds__(256) qkv_pack16_kernel(const T* __restrict__ qkv, float qscale, T* __restrict__ qh, T* __restrict__ kh,
                                                         T* __restrict__ vh, int Tn, int Tp, int H) {
  typedef T V8 __attribute__((ext_vector_type(8)));
  __shared__ __attribute__((aligned(16))) T sv[64][72];      // [token][channel], 144-B rows
  const int bh = blockIdx.x, b = bh / H, hd = bh % H;
  const int t0 = blockIdx.y * 64;
  const int HD = H * 64;
  const int tl = threadIdx.x >> 2, part = threadIdx.x & 3;   // token of the tile, 16-channel part
  const int t = t0 + tl;
  V8 q[2], k[2], v[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int u = 0; u < 8; ++u) { q[i][u] = (T)0.f; k[i][u] = (T)0.f; v[i][u] = (T)0.f; }
  if (t < Tn) {
    const T* p = qkv + ((long)b * Tn + t) * (3 * HD) + hd * 64 + part * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const V8 qr = *reinterpret_cast<const V8*>(p + 8 * i);
      k[i] = *reinterpret_cast<const V8*>(p + HD + 8 * i);
      v[i] = *reinterpret_cast<const V8*>(p + 2 * HD + 8 * i);
#pragma unroll
      for (int u = 0; u < 8; ++u) q[i][u] = (T)((float)qr[u] * qscale);
    }
  }
  const long o = ((long)bh * Tp + t) * 64 + part * 16;       // t < Tp always (Tp is a multiple of the 64-token tile); rows >= Tn are zero
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    *reinterpret_cast<V8*>(qh + o + 8 * i) = q[i];
    *reinterpret_cast<V8*>(kh + o + 8 * i) = k[i];
    *reinterpret_cast<V8*>(&sv[tl][part * 16 + 8 * i]) = v[i];
  }
  __syncthreads();
  // V^T [bh][64 channels][Tp]: thread = (channel, 8-token chunk), 512 items
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int item = threadIdx.x + 256 * it;
    const int d = item >> 3, ch = item & 7;
    V8 w;
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = sv[ch * 8 + u][d];
    *reinterpret_cast<V8*>(vh + ((long)bh * 64 + d) * Tp + t0 + ch * 8) = w;
  }
}

extern "C" int stedm_qkv_pack(const void* qkv, int qkv_is16, float qscale, void* q_hi, void* q_lo, void* k_hi, void* k_lo, void* vt_hi,
                              void* vt_lo, int B, int T, int Tp, int heads, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(qkv && q_hi && k_hi && vt_hi, "qkv_pack: null pointer");
  STEDM_CHECK_ARG(Tp % 128 == 0 && Tp >= T, "qkv_pack: Tp must be a multiple of 128 and >= T");
  STEDM_CHECK_ARG(!qkv_is16 || (!q_lo && !k_lo && !vt_lo), "qkv_pack: a 16-bit qkv input belongs to the single-product modes (no lo planes)");
  dim3 grid(B * heads, Tp / 64);
  if (qkv_is16) {
    if (mm_dtype == STEDM_F16)
      qkv_pack16_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>((const _Float16*)qkv, qscale, (_Float16*)q_hi, (_Float16*)k_hi, (_Float16*)vt_hi, T, Tp, heads);
    else
      qkv_pack16_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>((const __bf16*)qkv, qscale, (__bf16*)q_hi, (__bf16*)k_hi, (__bf16*)vt_hi, T, Tp, heads);
    STEDM_LAUNCH_CHECK();
    return 0;
  }
  const float* qkvf = reinterpret_cast<const float*>(qkv);
  if (mm_dtype == STEDM_F16)
    qkv_pack_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>(qkvf, qscale, (_Float16*)q_hi, (_Float16*)q_lo, (_Float16*)k_hi,
                                                                  (_Float16*)k_lo, (_Float16*)vt_hi, (_Float16*)vt_lo, T, Tp, heads);
  else
    qkv_pack_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>(qkvf, qscale, (__bf16*)q_hi, (__bf16*)q_lo, (__bf16*)k_hi, (__bf16*)k_lo,
                                                                (__bf16*)vt_hi, (__bf16*)vt_lo, T, Tp, heads);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ LSA flash attention
// One block = 128 queries of one (sample, head): 4 waves x 32 queries. Keys/values stream in tiles of 64.
// S^T = K Q^T is computed "swapped" (keys on the accumulator rows, the wave's 32 queries on the lanes), so that the
// row statistics of a query are lane-local and the exponentiated tile is directly the B operand of O^T += V^T P^T
// (cdna guide: "an accumulator tile as the next MFMA's operand"; its k order is 16s + 8(j>>2) + 4h + (j&3)).
struct FlashArgs {
  const void *qh, *ql, *kh, *kl, *vh, *vl;
  void *oh, *ol;   // [B][T][H*64]
  int T, Tp, H;
  // train-mode dropout of the attention probabilities (vit_set.py:43, 62; DROP kernels only): keep iff u16 >= thr16, kept scaled by inv_keep
  unsigned thr16, site;
  unsigned long long seed;
  float inv_keep;
};

template <typename T, int NPASS, bool DROP = false>
__global__ void __launch_bounds__(256, 2) lsa_flash_kernel(FlashArgs a) {
  using V8 = typename MM<T>::V8;
  typedef T V4t __attribute__((ext_vector_type(4)));
  constexpr int NPL = NPASS == 3 ? 2 : 1;
  constexpr int RS = 72;                          // K rows: LDS stride in elements (64 + 8 pad -> 144 B: conflict-free ds_read_b128 fragments)
  // V^T rows are read 8 B per lane by 32 consecutive rows (ds_read_b64: lane groups {0-31}, {32-63}): 144 B = 36 dwords repeats its bank
  // pair every 16 rows (2-way conflicts on every read, a third of the LDS cycles of the kernel: SQ_LDS_BANK_CONFLICT); 152 B = 38 dwords
  // gives 32 distinct pairs. Rows are then only 8-B aligned: the tile is stored with 8-B writes.
  constexpr int RSV = 76;
  // one LDS block: K and V^T tiles during the loop, the O^T transpose buffer afterwards
  constexpr int TILE = 64 * RSV;
  constexpr int KV_BYTES = 2 * NPL * TILE * (int)sizeof(T);
  constexpr int O_BYTES = 4 * 32 * 65 * (int)sizeof(float);
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[KV_BYTES > O_BYTES ? KV_BYTES : O_BYTES];
  T (*sK)[TILE] = reinterpret_cast<T (*)[TILE]>(lds_raw);
  T (*sV)[TILE] = reinterpret_cast<T (*)[TILE]>(lds_raw + NPL * TILE * sizeof(T));
  float (*sO)[32][65] = reinterpret_cast<float (*)[32][65]>(lds_raw);
  // XCD-aware block -> (sample-head, query tile): see lsa_flash64_kernel
  int bh, qtile;
  {
    const int nbh = gridDim.x, nq = gridDim.y, L = blockIdx.x + nbh * blockIdx.y;      // dispatch order of the 2-D grid
    if ((nbh & 7) == 0) { const int x = L & 7, j = L >> 3; bh = x + 8 * (j / nq); qtile = j % nq; }
    else { bh = blockIdx.x; qtile = blockIdx.y; }
  }
  const int b = bh / a.H, hd = bh % a.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int q0 = qtile * 128 + wave * 32;
  const T* qg[2] = {reinterpret_cast<const T*>(a.qh), reinterpret_cast<const T*>(a.ql)};
  const T* kg[2] = {reinterpret_cast<const T*>(a.kh), reinterpret_cast<const T*>(a.kl)};
  const T* vg[2] = {reinterpret_cast<const T*>(a.vh), reinterpret_cast<const T*>(a.vl)};

  V8 qf[NPL][4];
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qf[pl][ks] = *reinterpret_cast<const V8*>(qg[pl] + ((long)bh * a.Tp + q0 + r) * 64 + ks * 16 + h * 8);

  f32x16 o[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int qidx = q0 + r;
  U4 dstate{};
  if (DROP) dstate = attn_stream_init((unsigned)qidx, (unsigned)bh, a.site, (unsigned)h, a.seed);

  const int ntiles = (a.T + 63) / 64;
  // K / V^T tiles travel global -> registers -> LDS with the loads of tile kt + 1 issued BEFORE the MFMAs of tile kt (async-stage
  // split, cdna_hip_programming.md T14): their latency hides behind the tile's compute instead of sitting between two barriers
  // (named registers, macro-expanded: an array captured by a lambda went to scratch)
  uint4 kr0h, kr1h, vr0h, vr1h, kr0l, kr1l, vr0l, vr1l;
  const int frow0 = tid >> 3, frow1 = (tid + 256) >> 3, fc = tid & 7;
#define LSA_FETCH(KT)                                                                                                   \
  {                                                                                                                     \
    const long kb_ = ((long)bh * a.Tp + (KT) * 64) * 64 + fc * 8;                                                       \
    const long vb_ = (long)bh * 64 * a.Tp + (KT) * 64 + fc * 8;                                                         \
    kr0h = *reinterpret_cast<const uint4*>(kg[0] + kb_ + frow0 * 64);                                                   \
    kr1h = *reinterpret_cast<const uint4*>(kg[0] + kb_ + frow1 * 64);                                                   \
    vr0h = *reinterpret_cast<const uint4*>(vg[0] + vb_ + (long)frow0 * a.Tp);                                           \
    vr1h = *reinterpret_cast<const uint4*>(vg[0] + vb_ + (long)frow1 * a.Tp);                                           \
    if (NPASS == 3) {                                                                                                   \
      kr0l = *reinterpret_cast<const uint4*>(kg[1] + kb_ + frow0 * 64);                                                 \
      kr1l = *reinterpret_cast<const uint4*>(kg[1] + kb_ + frow1 * 64);                                                 \
      vr0l = *reinterpret_cast<const uint4*>(vg[1] + vb_ + (long)frow0 * a.Tp);                                         \
      vr1l = *reinterpret_cast<const uint4*>(vg[1] + vb_ + (long)frow1 * a.Tp);                                         \
    }                                                                                                                   \
  }
  LSA_FETCH(0)
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();   // every wave is done reading the previous tile
    *reinterpret_cast<uint4*>(&sK[0][frow0 * RS + fc * 8]) = kr0h;
    *reinterpret_cast<uint4*>(&sK[0][frow1 * RS + fc * 8]) = kr1h;
    *reinterpret_cast<uint2*>(&sV[0][frow0 * RSV + fc * 8]) = make_uint2(vr0h.x, vr0h.y);
    *reinterpret_cast<uint2*>(&sV[0][frow0 * RSV + fc * 8 + 4]) = make_uint2(vr0h.z, vr0h.w);
    *reinterpret_cast<uint2*>(&sV[0][frow1 * RSV + fc * 8]) = make_uint2(vr1h.x, vr1h.y);
    *reinterpret_cast<uint2*>(&sV[0][frow1 * RSV + fc * 8 + 4]) = make_uint2(vr1h.z, vr1h.w);
    if (NPASS == 3) {
      *reinterpret_cast<uint4*>(&sK[NPL - 1][frow0 * RS + fc * 8]) = kr0l;
      *reinterpret_cast<uint4*>(&sK[NPL - 1][frow1 * RS + fc * 8]) = kr1l;
      *reinterpret_cast<uint2*>(&sV[NPL - 1][frow0 * RSV + fc * 8]) = make_uint2(vr0l.x, vr0l.y);
      *reinterpret_cast<uint2*>(&sV[NPL - 1][frow0 * RSV + fc * 8 + 4]) = make_uint2(vr0l.z, vr0l.w);
      *reinterpret_cast<uint2*>(&sV[NPL - 1][frow1 * RSV + fc * 8]) = make_uint2(vr1l.x, vr1l.y);
      *reinterpret_cast<uint2*>(&sV[NPL - 1][frow1 * RSV + fc * 8 + 4]) = make_uint2(vr1l.z, vr1l.w);
    }
    __syncthreads();
    if (kt + 1 < ntiles) LSA_FETCH(kt + 1)
    // ---- S^T tiles (2 x 32 keys) x 32 queries
    f32x16 s[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[sub][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const V8 kf = *reinterpret_cast<const V8*>(&sK[0][(sub * 32 + r) * RS + ks * 16 + h * 8]);
        if (NPASS == 3) {
          const V8 kfl = *reinterpret_cast<const V8*>(&sK[NPL - 1][(sub * 32 + r) * RS + ks * 16 + h * 8]);
          s[sub] = MM<T>::mfma(kfl, qf[0][ks], s[sub]);
          s[sub] = MM<T>::mfma(kf, qf[NPL - 1][ks], s[sub]);
        }
        s[sub] = MM<T>::mfma(kf, qf[0][ks], s[sub]);
      }
    }
    // ---- mask (diagonal: a token never attends to itself, vit_set.py:58-60; padding keys) + online softmax. The logits arrive in
    // the log2 domain (qkv_pack folds log2(e) into q), so the exponential is the hardware exp2; the masks only touch the one tile
    // that holds this wave's own keys and the tiles past T (wave-uniform branches); the running output is rescaled only when some
    // lane's maximum actually moved.
    if ((q0 >> 6) == kt) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (kt * 64 + sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * h == qidx) s[sub][e] = -FLT_MAX;
    }
    if (kt * 64 + 64 > a.T) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (kt * 64 + sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * h >= a.T) s[sub][e] = -INFINITY;
    }
    float mx = s[0][0];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[sub][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float rs = 0.f;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __builtin_amdgcn_exp2f(s[sub][e] - m_new);
        s[sub][e] = pv;
        rs += pv;
      }
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
    m_run = m_new;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
    }
    if (DROP) {   // dropout acts on the normalised probabilities (vit_set.py:61-62): l keeps every p, the PV product only the kept ones
      const unsigned keep = attn_keep_bits(dstate, a.thr16);
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (!((keep >> (sub * 16 + e)) & 1u)) s[sub][e] = 0.f;
    }
    // ---- O^T += V^T P^T
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        V8 ph, pl8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float pv = s[sub][8 * s2 + j];
          const T hv = (T)pv;
          ph[j] = hv;
          if (NPASS == 3) pl8[j] = (T)(pv - (float)hv);
        }
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          // A fragment of V^T: row = d*32 + r, keys 32*sub + 16*s2 + {4h .. 4h+3} and {8 + 4h .. 8 + 4h + 3}
          const int base = (d * 32 + r) * RSV + sub * 32 + s2 * 16 + h * 4;
          V8 vf, vfl;
          const V4t v0 = *reinterpret_cast<const V4t*>(&sV[0][base]);
          const V4t v1 = *reinterpret_cast<const V4t*>(&sV[0][base + 8]);
          vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
          vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
          if (NPASS == 3) {
            const V4t w0 = *reinterpret_cast<const V4t*>(&sV[NPL - 1][base]);
            const V4t w1 = *reinterpret_cast<const V4t*>(&sV[NPL - 1][base + 8]);
            vfl[0] = w0[0]; vfl[1] = w0[1]; vfl[2] = w0[2]; vfl[3] = w0[3];
            vfl[4] = w1[0]; vfl[5] = w1[1]; vfl[6] = w1[2]; vfl[7] = w1[3];
            o[d] = MM<T>::mfma(vfl, ph, o[d]);
            o[d] = MM<T>::mfma(vf, pl8, o[d]);
          }
          o[d] = MM<T>::mfma(vf, ph, o[d]);
        }
      }
  }
#undef LSA_FETCH
  // ---- epilogue: O^T / l through LDS so that every token row is written contiguously (token-major [B][T][H*64])
  __syncthreads();   // all waves are done with the K / V^T tiles: the block is reused for the transpose
  const float inv = (DROP ? a.inv_keep : 1.0f) / l_run;
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) sO[wave][r][d * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] = o[d][e] * inv;
  __syncthreads();
  {
    const int row = lane >> 1, half = lane & 1;
    const int t = q0 + row;
    if (t < a.T) {
      T* oh = reinterpret_cast<T*>(a.oh) + ((long)b * a.T + t) * (a.H * 64) + hd * 64 + half * 32;
      T* ol = a.ol ? reinterpret_cast<T*>(a.ol) + ((long)b * a.T + t) * (a.H * 64) + hd * 64 + half * 32 : nullptr;
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const float v = sO[wave][row][half * 32 + j];
        const T hv = (T)v;
        oh[j] = hv;
        if (ol) ol[j] = (T)(v - (float)hv);
      }
    }
  }
}

// Single-product form, 64 queries per wave (round 3; replaces round 2's lsa_flash_dma_kernel, 32 queries per wave).
// K / V^T tiles travel global -> LDS by DMA (global_load_lds_dwordx4) into a ring of three tiles, two tiles ahead of the MFMAs, with ONE
// raw s_barrier per tile behind a counted vmcnt (a __syncthreads() makes the compiler wait for vmcnt(0) first, i.e. for the DMA of the NEXT
// tile: found in the ISA in round 2 — the ring had never run ahead). A DMA instruction writes its 64 x 16 B linearly, so rows are unpadded
// (128 B) and the 16-B pieces are XOR-swizzled through the SOURCE address: physical piece p of row w holds logical piece p ^ ((w >> 1) & 7) —
// conflict-free ds_read_b128 K fragments, 2-way on the 8-B V^T reads (measured 1 %).
// Blocks go to the 8 XCDs round-robin (block id mod 8) and each XCD has its own 4 MB L2: the blocks of one sample-head all run on one XCD,
// so the ~3 heads resident there fit its L2 (with the sample-head on the fast grid axis every XCD streamed the K / V^T of ALL heads: 3.4 GB
// of fabric traffic per call at B = 8).
// Softmax without a per-element subtraction: the logits leave the MFMA already relative to a per-query reference m_ref — a fifth k-step
// whose K side is the constant 1 and whose Q side is -m_ref — so p = exp2(s').
// Round 2's kernel gave a wave 32 queries, so every wave of the block read the whole K and V^T tile from LDS for 16 + 2 MFMAs: 16 KB of
// fragment reads per 576 MFMA cycles and wave, 12 waves per CU — the LDS pipe (128 B/clk per CU) as loaded as the matrix pipe — and its
// softmax (32 v_exp + ~60 other vector instructions) ran strictly between its QK^T and PV products: 1578 cycles per wave and tile against
// 576 of MFMA (0.78 PFLOP/s at B = 8, 0.87 at B = 64).
// Here a wave owns TWO 32-query blocks: a K / V^T fragment is read once into registers and feeds both blocks' MFMAs (half the LDS bytes per
// MFMA), and the two blocks give the in-order wave something to overlap with itself — the exponentials of block 0 are issued in the gaps
// of block 1's QK^T MFMAs, those of block 1 in the gaps of block 0's PV MFMAs (one MFMA, then a slice of 3 - 4 exponentials + their sums /
// conversions, fenced by sched_barrier). Two waves per SIMD (247 registers), 48 KB ring. 0.98 PFLOP/s at B = 64.
template <typename T, bool DROP = false>
__global__ void __launch_bounds__(256, 2) lsa_flash64_kernel(FlashArgs a) {
  using V8 = typename MM<T>::V8;
  typedef T V4t __attribute__((ext_vector_type(4)));
  constexpr int NBUF = 3, TILE_B = 16384;          // K 8 KiB | V^T 8 KiB
  constexpr int O_BYTES = 4 * 32 * 65 * (int)sizeof(float);
  constexpr int LDS_B = NBUF * TILE_B > O_BYTES ? NBUF * TILE_B : O_BYTES;
  __shared__ __attribute__((aligned(1024))) unsigned char ring[LDS_B];
  float (*sO)[32][65] = reinterpret_cast<float (*)[32][65]>(ring);
  // block -> (sample-head, 256-query tile), all tiles of a sample-head on one XCD
  int bh, qtile;
  {
    const int nq = (a.Tp + 255) / 256, nbh = gridDim.x / nq, L = blockIdx.x;
    if ((nbh & 7) == 0) { const int x = L & 7, j = L >> 3; bh = x + 8 * (j / nq); qtile = j % nq; }
    else { bh = L % nbh; qtile = L / nbh; }
  }
  const int b = bh / a.H, hd = bh % a.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int qw = qtile * 256 + wave * 64;           // first query of this wave; block qb covers qw + 32 qb ..
  const bool live = qw < a.T;                       // wave-uniform
  const T* qg = reinterpret_cast<const T*>(a.qh);
  const T* kg = reinterpret_cast<const T*>(a.kh);
  const T* vg = reinterpret_cast<const T*>(a.vh);

  V8 qf[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    int row = qw + qb * 32 + r;
    row = row < a.Tp ? row : a.Tp - 1;              // (Tp % 256 == 128: the last block's upper waves hold no queries; never stored)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *reinterpret_cast<const V8*>(qg + ((long)bh * a.Tp + row) * 64 + ks * 16 + h * 8);
  }
  const T* const kbase = kg + (long)bh * a.Tp * 64;
  const T* const vbase = vg + (long)bh * 64 * a.Tp;
  unsigned koff[2], voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * (2 * wave + i) + (lane >> 3);
    const int q = (lane & 7) ^ ((row >> 1) & 7);
    koff[i] = row * 64 + q * 8;
    voff[i] = row * a.Tp + q * 8;
  }
  auto issue = [&](int kt) __attribute__((always_inline)) {
    unsigned char* dst = ring + (kt % NBUF) * TILE_B + (2 * wave) * 1024;
    const T* kt_k = kbase + (long)kt * 4096;
    const T* kt_v = vbase + (long)kt * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      GLDS16(kt_k + koff[i], dst + i * 1024);
      GLDS16(kt_v + voff[i], dst + 8192 + i * 1024);
    }
  };
  f32x16 o[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[qb][d][e] = 0.f;
  float l_run[2] = {0.f, 0.f}, m_ref[2] = {0.f, 0.f};
  const int ntiles = (a.T + 63) / 64;
  U4 dstate[2] = {};
  if (DROP) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) dstate[qb] = attn_stream_init((unsigned)(qw + qb * 32 + r), (unsigned)bh, a.site, (unsigned)h, a.seed);
  }
  unsigned kb[2], vb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = i * 32 + r;
    kb[i] = row * 128 + ((h ^ ((row >> 1) & 7)) << 4);
    vb[i] = 8192 + row * 128 + (((row >> 1) & 7) << 4) + 8 * h;
  }
  V8 ka, qa[2];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ka[j] = (T)0.f; qa[0][j] = (T)0.f; qa[1][j] = (T)0.f; }
  if (h == 0) ka[0] = (T)1.f;

  // masks of the rare tiles: the block's own keys (diagonal, vit_set.py:58-60) and the keys beyond T
  auto masks = [&](f32x16 (&s)[2], const int qb, const int kt) __attribute__((always_inline)) {
    const int q0 = qw + qb * 32, qidx = q0 + r;
    if ((q0 >> 6) == kt) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (kt * 64 + sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * h == qidx) s[sub][e] = -FLT_MAX;
    }
    if (kt * 64 + 64 > a.T) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (kt * 64 + sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * h >= a.T) s[sub][e] = -INFINITY;
    }
  };
  // move of the reference m_ref to (at least) the block's maximum in this tile: O and l go to the new reference, the logits take the
  // difference explicitly. Runs on the first tile and in the rare redo below — NOT per tile: the common tile has
  // no row maximum at all (32 v_max3 + a cross-half shuffle + a ballot per block were a quarter of the softmax's vector instructions)
  auto move_ref = [&](f32x16 (&s)[2], const int qb, const bool first) __attribute__((always_inline)) {
    float mx = s[0][0];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[sub][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float up = first ? fmaxf(mx, -30000.f) : fmaxf(mx, 0.f);
    const float m_new = (float)(T)(m_ref[qb] + up);
    const float delta = m_new - m_ref[qb];
    const float alpha = __builtin_amdgcn_exp2f(-delta);
    m_ref[qb] = m_new;
    if (h == 0) qa[qb][0] = (T)(-m_new);
    l_run[qb] *= alpha;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[qb][d][e] *= alpha;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[sub][e] -= delta;
  };
  // elements [E0, E1) of a block's 32 logits: p = exp2(s'), row sum, and every completed octet packed to the operand type
  // (dropout: kept probabilities only; l keeps every p — vit_set.py:61-62)
#define SM_SLICE(S, QB, E0, E1, PH, RS2, KEEP)                                                              \
  _Pragma("unroll") for (int e_ = (E0); e_ < (E1); ++e_) {                                                 \
    const float pv_ = __builtin_amdgcn_exp2f(S[e_ >> 4][e_ & 15]);                                         \
    S[e_ >> 4][e_ & 15] = pv_;                                                                             \
    RS2[0] += pv_;   /* ONE chain: two chains are SLP-packed into v_pk_add_f32, which costs more beside MFMAs (and inline-asm adds   */ \
                     /* reading a v_exp result miss the transcendental-use wait state: measured wrong sums)                          */ \
    if ((e_ & 7) == 7) {                                                                                   \
      _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                                   \
        float v_ = S[e_ >> 4][(e_ & 15) - 7 + j_];                                                         \
        if (DROP && !(((KEEP) >> (e_ - 7 + j_)) & 1u)) v_ = 0.f;                                           \
        PH[e_ >> 3][j_] = (T)v_;                                                                           \
      }                                                                                                    \
    }                                                                                                      \
  }
  // Guard of the exponentials WITHOUT a row maximum: a tile's probabilities are bounded by their own row sum, which the loop has anyway.
  // If some query's sum exceeds kLimit (what the operand type — and with it the fp32 sums — can still carry: 2^14 for fp16, 2^80 for bf16;
  // an overflowed exponential makes the sum inf, a NaN fails the comparison too), the block's tile is redone from the K fragments in LDS
  // with the reference moved first.
  constexpr float kLimit = sizeof(typename MM<T>::V8) == 16 && __is_same(T, _Float16) ? 16384.f : 1.2089258e24f;
  auto redo = [&](const int qb, const int kt, const unsigned char* tb, V8 (&ph)[4], float (&rs2)[2], const unsigned keep) __attribute__((always_inline)) {
    f32x16 s[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[sub][e] = 0.f;
      s[sub] = MM<T>::mfma(ka, qa[qb], s[sub]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const V8 kfr = *reinterpret_cast<const V8*>(tb + (kb[sub] ^ (unsigned)(ks << 5)));
        s[sub] = MM<T>::mfma(kfr, qf[qb][ks], s[sub]);
      }
    }
    masks(s, qb, kt);
    move_ref(s, qb, false);
    rs2[0] = 0.f; rs2[1] = 0.f;
    SM_SLICE(s, qb, 0, 32, ph, rs2, keep)
  };

  issue(0);
  if (ntiles > 1) issue(1);
  for (int kt0 = 0; kt0 < ntiles; kt0 += NBUF) {
#pragma unroll
  for (int u = 0; u < NBUF; ++u) {
    const int kt = kt0 + u;
    if (kt >= ntiles) break;
    if (kt + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // tile kt is complete in LDS; every wave is done with tile kt - 1, whose buffer takes the next tile fetched
    if (kt + NBUF - 1 < ntiles) issue(kt + NBUF - 1);
    if (!live) continue;            // a wave without queries (the tail of the last 256-query tile: T = 4097 leaves three of its four waves
                                    // empty) only feeds the ring: no fragment reads, no MFMAs, no exponentials
    const unsigned char* tb = ring + u * TILE_B;
    // ---- K fragments: read once, used by both query blocks
    V8 kf[2][4];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[sub][ks] = *reinterpret_cast<const V8*>(tb + (kb[sub] ^ (unsigned)(ks << 5)));
    // ---- S^T of block 0
    f32x16 s0[2], s1[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s0[sub][e] = 0.f;
      s0[sub] = MM<T>::mfma(ka, qa[0], s0[sub]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s0[sub] = MM<T>::mfma(kf[sub][ks], qf[0][ks], s0[sub]);
    }
    masks(s0, 0, kt);
    if (kt == 0) move_ref(s0, 0, true);
    unsigned keep0 = ~0u, keep1 = ~0u;
    if (DROP) keep0 = attn_keep_bits(dstate[0], a.thr16);
    // ---- S^T of block 1 (10 MFMAs) with block 0's exponentials in the gaps
    V8 p0[4], p1[4];
    float rs0[2] = {0.f, 0.f}, rs1[2] = {0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 10; ++m) {
      const int sub = m / 5, km = m % 5;
      __builtin_amdgcn_sched_barrier(0);
      if (km == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) s1[sub][e] = 0.f;
        s1[sub] = MM<T>::mfma(ka, qa[1], s1[sub]);
      } else s1[sub] = MM<T>::mfma(kf[sub][km - 1], qf[1][km - 1], s1[sub]);
      __builtin_amdgcn_sched_barrier(0);
      SM_SLICE(s0, 0, (m * 32) / 10, ((m + 1) * 32) / 10, p0, rs0, keep0)
    }
    __builtin_amdgcn_sched_barrier(0);
    if (__builtin_amdgcn_ballot_w64(!(rs0[0] + rs0[1] <= kLimit))) redo(0, kt, tb, p0, rs0, keep0);
    l_run[0] += rs0[0] + rs0[1];
    // ---- V^T fragments: read once, used by both query blocks (fragment f = 4 sub + 2 s2 + d)
    V8 vf[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const unsigned vo = vb[f & 1] ^ (unsigned)((f >> 1) << 5);      // (4 sub + 2 s2) << 4 = (f >> 1) << 5
      const V4t v0 = *reinterpret_cast<const V4t*>(tb + vo);
      const V4t v1 = *reinterpret_cast<const V4t*>(tb + (vo ^ 16u));
      vf[f][0] = v0[0]; vf[f][1] = v0[1]; vf[f][2] = v0[2]; vf[f][3] = v0[3];
      vf[f][4] = v1[0]; vf[f][5] = v1[1]; vf[f][6] = v1[2]; vf[f][7] = v1[3];
    }
    masks(s1, 1, kt);
    if (kt == 0) move_ref(s1, 1, true);
    if (DROP) keep1 = attn_keep_bits(dstate[1], a.thr16);
    // ---- O^T of block 0 (8 MFMAs) with block 1's exponentials in the gaps
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      __builtin_amdgcn_sched_barrier(0);
      o[0][f & 1] = MM<T>::mfma(vf[f], p0[f >> 1], o[0][f & 1]);
      __builtin_amdgcn_sched_barrier(0);
      SM_SLICE(s1, 1, f * 4, f * 4 + 4, p1, rs1, keep1)
    }
    __builtin_amdgcn_sched_barrier(0);
    if (__builtin_amdgcn_ballot_w64(!(rs1[0] + rs1[1] <= kLimit))) redo(1, kt, tb, p1, rs1, keep1);
    l_run[1] += rs1[0] + rs1[1];
    // ---- O^T of block 1
#pragma unroll
    for (int f = 0; f < 8; ++f) o[1][f & 1] = MM<T>::mfma(vf[f], p1[f >> 1], o[1][f & 1]);
  }
  }
#undef SM_SLICE
  // ---- epilogue: O^T / l through LDS, one query block after the other, every token row written contiguously ([B][T][H*64])
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float lr = l_run[qb];
    lr += __shfl_xor(lr, 32, 64);
    const float inv = live ? (DROP ? a.inv_keep : 1.0f) / lr : 0.f;
    __syncthreads();   // all waves are done with the ring (qb = 1: with the previous block's rows)
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) sO[wave][r][d * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] = o[qb][d][e] * inv;
    __syncthreads();
    const int row = lane >> 1, half = lane & 1;
    const int t = qw + qb * 32 + row;
    if (t < a.T) {
      T* oh = reinterpret_cast<T*>(a.oh) + ((long)b * a.T + t) * (a.H * 64) + hd * 64 + half * 32;
#pragma unroll
      for (int j = 0; j < 32; ++j) oh[j] = (T)sO[wave][row][half * 32 + j];
    }
  }
}

static int lsa_flash_launch(FlashArgs a, int B, int npass, int mm_dtype, bool drop, hipStream_t st) {
  dim3 grid(B * a.H, a.Tp / 128);
  if (npass == 1) {
    const dim3 grid64(B * a.H * ((a.Tp + 255) / 256));
    if (mm_dtype == STEDM_F16) { if (drop) lsa_flash64_kernel<_Float16, true><<<grid64, 256, 0, st>>>(a); else lsa_flash64_kernel<_Float16><<<grid64, 256, 0, st>>>(a); }
    else { if (drop) lsa_flash64_kernel<__bf16, true><<<grid64, 256, 0, st>>>(a); else lsa_flash64_kernel<__bf16><<<grid64, 256, 0, st>>>(a); }
    STEDM_LAUNCH_CHECK();
    return 0;
  }
  if (mm_dtype == STEDM_F16) {
    if (npass == 3) { if (drop) lsa_flash_kernel<_Float16, 3, true><<<grid, 256, 0, st>>>(a); else lsa_flash_kernel<_Float16, 3><<<grid, 256, 0, st>>>(a); }
    else { if (drop) lsa_flash_kernel<_Float16, 1, true><<<grid, 256, 0, st>>>(a); else lsa_flash_kernel<_Float16, 1><<<grid, 256, 0, st>>>(a); }
  } else {
    if (npass == 3) { if (drop) lsa_flash_kernel<__bf16, 3, true><<<grid, 256, 0, st>>>(a); else lsa_flash_kernel<__bf16, 3><<<grid, 256, 0, st>>>(a); }
    else { if (drop) lsa_flash_kernel<__bf16, 1, true><<<grid, 256, 0, st>>>(a); else lsa_flash_kernel<__bf16, 1><<<grid, 256, 0, st>>>(a); }
  }
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_lsa_flash(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, const void* vt_hi,
                               const void* vt_lo, void* out_hi, void* out_lo, int B, int T, int Tp, int heads, int npass,
                               int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(q_hi && k_hi && vt_hi && out_hi, "lsa_flash: null pointer");
  STEDM_CHECK_ARG(npass == 1 || (q_lo && k_lo && vt_lo && out_lo), "lsa_flash: npass=3 needs lo planes");
  STEDM_CHECK_ARG(Tp % 128 == 0 && Tp >= T && T > 1, "lsa_flash: need Tp %% 128 == 0, Tp >= T > 1");
  FlashArgs a{q_hi, q_lo, k_hi, k_lo, vt_hi, vt_lo, out_hi, npass == 3 ? out_lo : nullptr, T, Tp, heads, 0u, 0u, 0ull, 1.0f};
  return lsa_flash_launch(a, B, npass, mm_dtype, false, as_stream(stream));
}

extern "C" int stedm_lsa_flash_drop(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, const void* vt_hi,
                                    const void* vt_lo, void* out_hi, void* out_lo, int B, int T, int Tp, int heads, int npass,
                                    int mm_dtype, float p, unsigned long long seed, unsigned site, void* stream) {
  STEDM_CHECK_ARG(q_hi && k_hi && vt_hi && out_hi, "lsa_flash_drop: null pointer");
  STEDM_CHECK_ARG(npass == 1 || (q_lo && k_lo && vt_lo && out_lo), "lsa_flash_drop: npass=3 needs lo planes");
  STEDM_CHECK_ARG(Tp % 128 == 0 && Tp >= T && T > 1, "lsa_flash_drop: need Tp %% 128 == 0, Tp >= T > 1");
  STEDM_CHECK_ARG(p >= 0.f && p < 1.f, "lsa_flash_drop: p must be in [0, 1)");
  const unsigned thr = (unsigned)lrint((double)p * 65536.0);
  STEDM_CHECK_ARG(thr <= 65535u, "lsa_flash_drop: p too close to 1");
  FlashArgs a{q_hi, q_lo, k_hi, k_lo, vt_hi, vt_lo, out_hi, npass == 3 ? out_lo : nullptr, T, Tp, heads, thr, site, seed, 1.0f / (1.0f - p)};
  return lsa_flash_launch(a, B, npass, mm_dtype, true, as_stream(stream));
}

// ------------------------------------------------------------------------------------------------ train-mode dropout (elementwise sites)
// out = drop(src) (+ res), as fp32 and / or as 16-bit operand planes: the sites after pos_embedding (vit_set.py:187), after to_out's
// Linear (:49), after the FeedForward's GELU and its second Linear (:28-30). A thread owns 8 consecutive elements = one Philox call.
template <typename T>
__global__ void __launch_bounds__(256) dropout_rows_kernel(const float* src, const float* res, float* out,   // (out may alias src or res: in-place sites)
                                                           T* __restrict__ hi, T* __restrict__ lo, long n, unsigned thr16, float inv_keep,
                                                           unsigned long long seed, unsigned site) {
  const long ngroups = (n + 7) >> 3;
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (long)gridDim.x * 256) {
    const U4 r = drop_group((uint64_t)g, site, seed);
    const long e0 = g << 3;
    float v[8];
    if (e0 + 8 <= n) {
      const float4 a0 = reinterpret_cast<const float4*>(src + e0)[0], a1 = reinterpret_cast<const float4*>(src + e0)[1];
      v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = e0 + j < n ? src[e0 + j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = drop_u16(r, j) >= thr16 ? __fmul_rn(v[j], inv_keep) : 0.f;   // (rounded product: no fma with the residual)
    if (res) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (e0 + j < n) v[j] += res[e0 + j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (e0 + j >= n) break;
      if (out) out[e0 + j] = v[j];
      if (hi) {
        const T hv = (T)v[j];
        hi[e0 + j] = hv;
        if (lo) lo[e0 + j] = (T)(v[j] - (float)hv);
      }
    }
  }
}

extern "C" int stedm_dropout_rows(const float* src, const float* res, float* out, void* out_hi, void* out_lo, long n, float p,
                                  unsigned long long seed, unsigned site, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(src && (out || out_hi) && n > 0, "dropout_rows: bad args");
  STEDM_CHECK_ARG(!out_lo || out_hi, "dropout_rows: a lo plane needs its hi plane");
  STEDM_CHECK_ARG(p >= 0.f && p < 1.f, "dropout_rows: p must be in [0, 1)");
  STEDM_CHECK_ARG(((uintptr_t)src & 15) == 0, "dropout_rows: src must be 16-byte aligned");
  const unsigned thr = (unsigned)lrint((double)p * 65536.0);
  STEDM_CHECK_ARG(thr <= 65535u, "dropout_rows: p too close to 1");
  const long ngroups = (n + 7) >> 3;
  const int grid = (int)((ngroups + 255) / 256 < 32768 ? (ngroups + 255) / 256 : 32768);
  const float inv_keep = 1.0f / (1.0f - p);
  if (mm_dtype == STEDM_F16)
    dropout_rows_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>(src, res, out, (_Float16*)out_hi, (_Float16*)out_lo, n, thr, inv_keep, seed, site);
  else
    dropout_rows_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>(src, res, out, (__bf16*)out_hi, (__bf16*)out_lo, n, thr, inv_keep, seed, site);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ pooled head
// stage 1 of the token pooling when the batch alone cannot fill the chip: block (b, slab) sums its run of tokens (16-B loads, fixed order)
// into part[b][slab][dim]; the head kernel then pools the slabs
__global__ void __launch_bounds__(256) svit_pool_partial_kernel(const float* __restrict__ x, int Tn, int dim, int nslab, float* __restrict__ part) {
  extern __shared__ float spp[];                 // [lanes][dim]
  const int b = blockIdx.x, slab = blockIdx.y;
  const int per = (Tn + nslab - 1) / nslab;
  const int t0 = slab * per, t1 = min(Tn, t0 + per);
  const float* px = x + (long)b * Tn * dim;
  const int Q = dim >> 2, lanes = 256 / Q;
  const int q = threadIdx.x % Q, tl = threadIdx.x / Q;
  if (tl < lanes) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int t = t0 + tl; t < t1; t += lanes) {
      const float4 v = *reinterpret_cast<const float4*>(px + (long)t * dim + q * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(spp + (long)tl * dim + q * 4) = acc;
  }
  __syncthreads();
  for (int n = threadIdx.x; n < dim; n += 256) {
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += spp[(long)l * dim + n];
    part[((long)b * nslab + slab) * dim + n] = s;
  }
}

// Tn tokens (or slab partials) of x are pooled; the mean divides by `mean_n` (the real token count)
__global__ void __launch_bounds__(256) svit_head_kernel(const float* __restrict__ x, int Tn, int mean_n, int dim, int pool,
                                                        const float* __restrict__ c_old, const float* __restrict__ g,
                                                        const float* __restrict__ bt, float eps, const float* __restrict__ wt,
                                                        const float* __restrict__ bias, float* __restrict__ out, int ncls) {
  extern __shared__ float sp[];   // [dim] pooled, then normalised
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float* px = x + (long)b * Tn * dim;
  if (pool != 1 && (dim & 3) == 0 && dim <= 1024) {
    // token pooling: 256 threads = Q channel quads x (256 / Q) token lanes, 16-B loads; the lanes meet in LDS in a fixed order
    float* spart = sp + dim;                      // [lanes][dim]
    const int Q = dim >> 2, lanes = 256 / Q;
    const int q = threadIdx.x % Q, tl = threadIdx.x / Q;
    if (tl < lanes) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
      for (int t = tl; t < Tn; t += lanes) {
        const float4 v = *reinterpret_cast<const float4*>(px + (long)t * dim + q * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      *reinterpret_cast<float4*>(spart + (long)tl * dim + q * 4) = acc;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < dim; n += 256) {
      float s = 0.f;
      for (int l = 0; l < lanes; ++l) s += spart[(long)l * dim + n];
      if (pool == 0) s /= (float)mean_n;
      if (c_old) s += c_old[(long)b * dim + n];
      sp[n] = s;
    }
  } else {
    for (int n = threadIdx.x; n < dim; n += 256) {
      float s = 0.f;
      if (pool == 1) s = px[n];
      else {
        for (int t = 0; t < Tn; ++t) s += px[(long)t * dim + n];
        if (pool == 0) s /= (float)mean_n;
      }
      if (c_old) s += c_old[(long)b * dim + n];
      sp[n] = s;
    }
  }
  __syncthreads();
  float s1 = 0.f;
  for (int n = threadIdx.x; n < dim; n += 256) s1 += sp[n];
  const float mean = block_sum_256(s1, red) / dim;
  float s2 = 0.f;
  for (int n = threadIdx.x; n < dim; n += 256) { const float d = sp[n] - mean; s2 += d * d; }
  const float rstd = 1.0f / sqrtf(block_sum_256(s2, red) / dim + eps);
  __syncthreads();
  for (int n = threadIdx.x; n < dim; n += 256) sp[n] = (sp[n] - mean) * rstd * g[n] + bt[n];
  __syncthreads();
  for (int n = threadIdx.x; n < ncls; n += 256) {
    float acc = bias[n];
    for (int k = 0; k < dim; ++k) acc = fmaf(sp[k], wt[(long)k * ncls + n], acc);
    out[(long)b * ncls + n] = acc;
  }
}

extern "C" int stedm_svit_head(const float* x, int B, int T, int dim, int pool, const float* c_old, const float* ln_w,
                               const float* ln_b, float eps, const float* wt, const float* bias, float* out, int ncls,
                               float* ws, long ws_floats, void* stream) {
  STEDM_CHECK_ARG(x && ln_w && ln_b && wt && bias && out, "svit_head: null pointer");
  STEDM_CHECK_ARG(pool >= 0 && pool <= 2, "svit_head: pool must be 0 (mean), 1 (cls) or 2 (sum)");
  const bool vec = dim % 4 == 0 && dim <= 1024;
  const int lanes = vec ? 256 / (dim / 4) : 0;
  const size_t lds = (size_t)(1 + lanes) * dim * sizeof(float);
  // token pooling over many blocks when B alone leaves the chip idle (B = 8: 8 of 256 CUs read 34 MB): slabs of >= 32 tokens, about
  // 1024 blocks in all, as many as the caller's workspace admits
  int nslab = 1;
  if (pool != 1 && vec && ws && T >= 256) {
    nslab = (1024 + B - 1) / B;
    if (nslab > T / 32) nslab = T / 32;
    if ((long)B * nslab * dim > ws_floats) nslab = (int)(ws_floats / ((long)B * dim));
  }
  if (nslab >= 2) {
    svit_pool_partial_kernel<<<dim3(B, nslab), 256, (size_t)lanes * dim * sizeof(float), as_stream(stream)>>>(x, T, dim, nslab, ws);
    STEDM_LAUNCH_CHECK();
    svit_head_kernel<<<B, 256, lds, as_stream(stream)>>>(ws, nslab, T, dim, pool, c_old, ln_w, ln_b, eps, wt, bias, out, ncls);
  } else {
    svit_head_kernel<<<B, 256, lds, as_stream(stream)>>>(x, T, T, dim, pool, c_old, ln_w, ln_b, eps, wt, bias, out, ncls);
  }
  STEDM_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ agg blocks / rescaler
__global__ void agg_reduce_kernel(const float* __restrict__ f, float* __restrict__ out, int n, int F, int mode, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long b = i / F;
  const int k = (int)(i - b * F);
  float acc = f[(b * n) * F + k];
  for (int j = 1; j < n; ++j) {
    const float v = f[(b * n + j) * F + k];
    acc = mode == 1 ? fmaxf(acc, v) : acc + v;
  }
  out[i] = mode == 1 ? acc : acc / (float)n;
}

extern "C" int stedm_agg_reduce(const float* feats, float* out, int B, int n, int F, int mode, void* stream) {
  STEDM_CHECK_ARG(feats && out && B > 0 && n > 0 && F > 0 && (mode == 0 || mode == 1), "agg_reduce: bad args");
  const long total = (long)B * F;
  agg_reduce_kernel<<<(int)((total + 255) / 256), 256, 0, as_stream(stream)>>>(feats, out, n, F, mode, total);
  STEDM_LAUNCH_CHECK();
  return 0;
}

__global__ void spatial_rescale_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, int cin,
                                       int cout, int H, int W, int f, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int Wo = W / f, Ho = H / f;
  const int xo = (int)(i % Wo);
  const int yo = (int)((i / Wo) % Ho);
  const int co = (int)((i / ((long)Wo * Ho)) % cout);
  const long b = i / ((long)Wo * Ho * cout);
  float acc = 0.f;
  for (int ci = 0; ci < cin; ++ci) {
    // n stages of 2x2 box means == one f x f box mean, but keep the staged summation order of the reference
    float s = 0.f;
    const float* p = x + ((b * cin + ci) * H + (long)yo * f) * W + (long)xo * f;
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx) s += p[(long)dy * W + dx];
    acc = fmaf(s / (float)(f * f), w ? w[co * cin + ci] : (ci == co ? 1.f : 0.f), acc);
  }
  out[i] = acc;
}

extern "C" int stedm_spatial_rescale(const float* x, const float* w, float* out, int B, int cin, int cout, int H, int W,
                                     int n_stages, void* stream) {
  STEDM_CHECK_ARG(x && out && n_stages >= 0 && n_stages < 8, "spatial_rescale: bad args");
  const int f = 1 << n_stages;
  STEDM_CHECK_ARG(H % f == 0 && W % f == 0, "spatial_rescale: H, W must be divisible by 2^n_stages (bilinear 1/2 == box mean only then)");
  STEDM_CHECK_ARG(w || cin == cout, "spatial_rescale: no channel mapper needs cin == cout");
  const long total = (long)B * cout * (H / f) * (W / f);
  spatial_rescale_kernel<<<(int)((total + 255) / 256), 256, 0, as_stream(stream)>>>(x, w, out, cin, cout, H, W, f, total);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ GEGLU
// attention.py:37-44: x, gate = proj(x).chunk(2, -1); out = x * gelu(gate) (exact erf GELU) -> 16-bit planes
template <typename T>
__global__ void __launch_bounds__(256) geglu16_kernel(const float* __restrict__ g, T* __restrict__ hi, T* __restrict__ lo, long M,
                                                      int I) {
  const long total = M * (I / 4);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / (I / 4);
    const int q = (int)(i - row * (I / 4));
    const float4 x = *reinterpret_cast<const float4*>(g + row * 2 * I + q * 4);
    const float4 t = *reinterpret_cast<const float4*>(g + row * 2 * I + I + q * 4);
    const float xs[4] = {x.x, x.y, x.z, x.w}, ts[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = xs[j] * gelu_erf_f(ts[j]);
      const T h = (T)v;
      hi[row * I + q * 4 + j] = h;
      if (lo) lo[row * I + q * 4 + j] = (T)(v - (float)h);
    }
  }
}

extern "C" int stedm_geglu16(const float* g, void* out_hi, void* out_lo, long M, int I, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(g && out_hi && M > 0 && I > 0 && I % 4 == 0, "geglu16: bad args");
  long blocks = (M * (I / 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (mm_dtype == STEDM_F16) geglu16_kernel<_Float16><<<(int)blocks, 256, 0, as_stream(stream)>>>(g, (_Float16*)out_hi, (_Float16*)out_lo, M, I);
  else geglu16_kernel<__bf16><<<(int)blocks, 256, 0, as_stream(stream)>>>(g, (__bf16*)out_hi, (__bf16*)out_lo, M, I);
  STEDM_LAUNCH_CHECK();
  return 0;
}
