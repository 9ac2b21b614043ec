// Swin-Transformer-V2 style embedder: the pieces of torchvision's `swin_v2_t` (pinned torchvision==0.18.1,
// call sites networks/s_zss_dm.py:19-20, networks/agg_blocks.py:28,49,70) that are not GEMMs. torchvision is a third-party dependency
// absent from /root/reference: the kernels restate its published algorithm (Liu et al., "Swin Transformer V2", and the layer
// definitions of torchvision.models.swin_transformer) — parity unpinned, see oracle/swin.py.
//
//   swin_patch16      : features[0][0] Conv2d(3, 96, 4, stride 4) as a GEMM: 4x4x3 patch gather -> 16-bit operand rows [tok][64]
//                       (k = c*16 + ky*4 + kx, the OIHW order of the conv weight; columns 48..63 zero), any input strides
//   swin_ln           : out = res + LayerNorm(y) (post-norm residual of SwinTransformerBlockV2; res NULL: the plain LayerNorms of
//                       features[0][2], PatchMergingV2.norm and the final norm) as fp32 rows and / or 16-bit operand planes
//   swin_window_attn  : shifted_window_attention with cosine logits (ShiftedWindowAttentionV2): cyclic shift + window partition as index
//                       arithmetic, F.normalize(q) . F.normalize(k) * exp(min(logit_scale, log 100)) + 16 sigmoid(cpb) + shift mask,
//                       softmax, P V, written back at the tokens' own positions as the 16-bit plane `proj` consumes. Windows that hang
//                       over the feature map (F.pad) see zero rows: q = q_bias, k = 0, v = v_bias, exactly as the padded Linear gives.
//   swin_merge16      : PatchMergingV2's 2x2 neighbourhood concat [x(0,0) | x(1,0) | x(0,1) | x(1,1)] -> 16-bit operand rows [tok][4C]
//   swin_token_mean   : AdaptiveAvgPool2d(1) over the tokens of an image
#include <float.h>

#include "common.hpp"
using namespace stedm;

namespace {

template <typename T>
__global__ void __launch_bounds__(256) swin_patch16_kernel(const float* __restrict__ img, long sn, long sc, long sh, long sw, T* __restrict__ hi,
                                                           T* __restrict__ lo, int Hp, int Wp, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i & 63);
  const long tok = i >> 6;
  float v = 0.f;
  if (k < 48) {
    const int tx = (int)(tok % Wp);
    const long r = tok / Wp;
    const int ty = (int)(r % Hp);
    const long n = r / Hp;
    const int c = k >> 4, ky = (k >> 2) & 3, kx = k & 3;
    v = img[n * sn + c * sc + (long)(4 * ty + ky) * sh + (long)(4 * tx + kx) * sw];
  }
  const T h = (T)v;
  hi[i] = h;
  if (lo) lo[i] = (T)(v - (float)h);
}

// one wave per row
template <typename T>
__global__ void __launch_bounds__(256) swin_ln_kernel(const float* __restrict__ y, const float* __restrict__ g, const float* __restrict__ bt,
                                                      float eps, const float* __restrict__ res, float* __restrict__ out, T* __restrict__ hi,
                                                      T* __restrict__ lo, long rows, int dim) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* py = y + row * dim;
  float s1 = 0.f;
  for (int k = lane; k < dim; k += 64) s1 += py[k];
  const float mean = wave_sum(s1) / dim;
  float s2 = 0.f;
  for (int k = lane; k < dim; k += 64) { const float d = py[k] - mean; s2 += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(s2) / dim + eps);
  for (int k = lane; k < dim; k += 64) {
    float v = (py[k] - mean) * rstd * g[k] + bt[k];
    if (res) v += res[row * dim + k];
    if (out) out[row * dim + k] = v;
    if (hi) {
      const T h = (T)v;
      hi[row * dim + k] = h;
      if (lo) lo[row * dim + k] = (T)(v - (float)h);
    }
  }
}

struct SwinAttnArgs {
  const float* qkv;     // [N*H*W][3C]  ('(three heads d)' columns, bias included)
  const float* bias;    // [3C] the Linear's bias with the k third zeroed (pad rows)
  const float* scale;   // [heads] exp(min(logit_scale, log 100))
  const float* rpbT;    // [heads][64 keys][64 queries] 16 * sigmoid(cpb_mlp(table))[index], key-major so that a wave reads rows
  void* hi;             // [N*H*W][C]
  void* lo;
  int H, W, C, heads, shift_h, shift_w, padH, padW;
};

// one wave per (window, head): lane = query token of the 8 x 8 window, head dim 32
template <typename T>
__global__ void __launch_bounds__(64) swin_window_attn_kernel(const SwinAttnArgs a) {
  __shared__ __attribute__((aligned(16))) float sk[64][32];
  __shared__ __attribute__((aligned(16))) float sv[64][32];
  __shared__ int sid[64];
  const int lane = threadIdx.x;
  const int nwx = a.padW >> 3;
  const int wy = blockIdx.x / nwx, wx = blockIdx.x % nwx;
  const int h = blockIdx.y;
  const long n = blockIdx.z;
  const int ys = wy * 8 + (lane >> 3), xs = wx * 8 + (lane & 7);        // position in the shifted (rolled) frame
  int y = ys + a.shift_h; if (y >= a.padH) y -= a.padH;                  // torch.roll(x, -shift): rolled[ys] = x[(ys + shift) % pad]
  int x = xs + a.shift_w; if (x >= a.padW) x -= a.padW;
  const bool valid = y < a.H && x < a.W;
  const long tok = (n * a.H + y) * a.W + x;
  const int C = a.C, c0 = h * 32;
  float q[32], o[32];
  {
    const float* pq = valid ? a.qkv + tok * (3L * C) + c0 : a.bias + c0;
    const float* pk = pq + C;
    const float* pv = pq + 2 * C;
    float kk[32];
    float nq = 0.f, nk = 0.f;
#pragma unroll
    for (int d = 0; d < 32; d += 4) {
      const float4 q4 = *reinterpret_cast<const float4*>(pq + d);
      const float4 k4 = *reinterpret_cast<const float4*>(pk + d);
      const float4 v4 = *reinterpret_cast<const float4*>(pv + d);
      q[d] = q4.x; q[d + 1] = q4.y; q[d + 2] = q4.z; q[d + 3] = q4.w;
      kk[d] = k4.x; kk[d + 1] = k4.y; kk[d + 2] = k4.z; kk[d + 3] = k4.w;
      *reinterpret_cast<float4*>(&sv[lane][d]) = v4;
    }
#pragma unroll
    for (int d = 0; d < 32; ++d) { nq += q[d] * q[d]; nk += kk[d] * kk[d]; }
    // F.normalize(p=2, eps=1e-12): v / max(||v||, eps)
    const float rq = 1.0f / fmaxf(sqrtf(nq), 1e-12f), rk = 1.0f / fmaxf(sqrtf(nk), 1e-12f);
#pragma unroll
    for (int d = 0; d < 32; ++d) { q[d] *= rq; kk[d] *= rk; }
#pragma unroll
    for (int d = 0; d < 32; d += 4) *reinterpret_cast<float4*>(&sk[lane][d]) = make_float4(kk[d], kk[d + 1], kk[d + 2], kk[d + 3]);
  }
  // shift mask regions of the rolled frame: rows [0, pad-ws) / [pad-ws, pad-shift) / [pad-shift, pad), same for columns
  int rid = 0;
  if (a.shift_h | a.shift_w) {
    // (an unshifted side is one region: torchvision's third slice [-0:] then covers, and overwrites, the whole side)
    const int ih = !a.shift_h ? 0 : (ys < a.padH - 8 ? 0 : (ys < a.padH - a.shift_h ? 1 : 2));
    const int iw = !a.shift_w ? 0 : (xs < a.padW - 8 ? 0 : (xs < a.padW - a.shift_w ? 1 : 2));
    rid = ih * 3 + iw;
  }
  sid[lane] = rid;
  __syncthreads();
  const float sc = a.scale[h];
  const float* rp = a.rpbT + (long)h * 4096 + lane;
  float s[64];
  float mx = -FLT_MAX;
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < 32; d += 4) {
      const float4 k4 = *reinterpret_cast<const float4*>(&sk[j][d]);
      acc += q[d] * k4.x; acc += q[d + 1] * k4.y; acc += q[d + 2] * k4.z; acc += q[d + 3] * k4.w;
    }
    float v = acc * sc + rp[j * 64];
    if (sid[j] != rid) v += -100.0f;
    s[j] = v;
    mx = fmaxf(mx, v);
  }
  float sum = 0.f;
#pragma unroll
  for (int d = 0; d < 32; ++d) o[d] = 0.f;
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    const float pj = __expf(s[j] - mx);
    sum += pj;
#pragma unroll
    for (int d = 0; d < 32; d += 4) {
      const float4 v4 = *reinterpret_cast<const float4*>(&sv[j][d]);
      o[d] += pj * v4.x; o[d + 1] += pj * v4.y; o[d + 2] += pj * v4.z; o[d + 3] += pj * v4.w;
    }
  }
  if (!valid) return;
  const float inv = 1.0f / sum;
  T* ph = reinterpret_cast<T*>(a.hi) + tok * C + c0;
  T* pl = a.lo ? reinterpret_cast<T*>(a.lo) + tok * C + c0 : nullptr;
  typedef T V8 __attribute__((ext_vector_type(8)));
#pragma unroll
  for (int d = 0; d < 32; d += 8) {
    V8 h8, l8;
#pragma unroll
    for (int u = 0; u < 8; ++u) { const float v = o[d + u] * inv; h8[u] = (T)v; l8[u] = (T)(v - (float)h8[u]); }
    *reinterpret_cast<V8*>(ph + d) = h8;
    if (pl) *reinterpret_cast<V8*>(pl + d) = l8;
  }
}

// one thread per 4 channels of an output row [tok][4C]
template <typename T>
__global__ void __launch_bounds__(256) swin_merge16_kernel(const float* __restrict__ x, T* __restrict__ hi, T* __restrict__ lo, int H, int W, int C,
                                                           int Ho, int Wo, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int Q = C >> 2;                 // channel quads per source pixel
  const int cq = (int)(i % Q);
  long r = i / Q;
  const int part = (int)(r & 3); r >>= 2;
  const int xo = (int)(r % Wo); r /= Wo;
  const int yo = (int)(r % Ho);
  const long n = r / Ho;
  const int y = 2 * yo + (part & 1), xx = 2 * xo + (part >> 1);     // x0 (0,0), x1 (1,0), x2 (0,1), x3 (1,1) as (dy, dx)
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (y < H && xx < W) v = *reinterpret_cast<const float4*>(x + ((n * H + y) * W + xx) * C + cq * 4);
  typedef T V4 __attribute__((ext_vector_type(4)));
  const float f[4] = {v.x, v.y, v.z, v.w};
  V4 h4, l4;
#pragma unroll
  for (int u = 0; u < 4; ++u) { h4[u] = (T)f[u]; l4[u] = (T)(f[u] - (float)h4[u]); }
  const long o = (((n * Ho + yo) * Wo + xo) * 4 + part) * C + cq * 4;
  *reinterpret_cast<V4*>(hi + o) = h4;
  if (lo) *reinterpret_cast<V4*>(lo + o) = l4;
}

// rpbT[h][key j][query i] = 16 sigmoid(cpb[index[i * 64 + j]][h]) (ShiftedWindowAttentionV2.get_relative_position_bias)
__global__ void __launch_bounds__(256) swin_rpb_kernel(const float* __restrict__ cpb, const long* __restrict__ index, float* __restrict__ out,
                                                       int heads, int ntab) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // over heads * 4096
  if (i >= heads * 4096) return;
  const int h = i >> 12, j = (i >> 6) & 63, q = i & 63;
  long e = index[q * 64 + j];
  e = e < 0 ? 0 : (e >= ntab ? ntab - 1 : e);
  out[i] = 16.0f / (1.0f + __expf(-cpb[e * heads + h]));
}

// mean over the T tokens of an image: x [N][T][C] -> out [N][C]; block = (image, 64-channel slab), 4 waves over the tokens, fixed order
__global__ void __launch_bounds__(256) swin_token_mean_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int C) {
  __shared__ float part[4][64];
  const int n = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C)
    for (int t = w; t < T; t += 4) s += x[((long)n * T + t) * C + c];
  part[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < C) out[(long)n * C + c] = (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]) / T;
}

}  // namespace

extern "C" int stedm_swin_patch16(const float* img, long sn, long sc, long sh, long sw, int N, int H, int W, void* out_hi, void* out_lo,
                                  int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(img && out_hi && N > 0 && H > 0 && W > 0, "swin_patch16: bad args");
  STEDM_CHECK_ARG(H % 4 == 0 && W % 4 == 0, "swin_patch16: image sides must be multiples of the 4-pixel patch (H=%d W=%d)", H, W);
  const int Hp = H / 4, Wp = W / 4;
  const long total = (long)N * Hp * Wp * 64;
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (mm_dtype == STEDM_F16)
    swin_patch16_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>(img, sn, sc, sh, sw, (_Float16*)out_hi, (_Float16*)out_lo, Hp, Wp, total);
  else
    swin_patch16_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>(img, sn, sc, sh, sw, (__bf16*)out_hi, (__bf16*)out_lo, Hp, Wp, total);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_swin_ln(const float* y, const float* gamma, const float* beta, float eps, const float* res, float* out, void* out_hi,
                             void* out_lo, long rows, int dim, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(y && gamma && beta && (out || out_hi) && rows > 0 && dim > 0, "swin_ln: bad args");
  STEDM_CHECK_ARG(!out_lo || out_hi, "swin_ln: out_lo without out_hi");
  const unsigned grid = (unsigned)((rows + 3) / 4);
  if (mm_dtype == STEDM_F16)
    swin_ln_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>(y, gamma, beta, eps, res, out, (_Float16*)out_hi, (_Float16*)out_lo, rows, dim);
  else
    swin_ln_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>(y, gamma, beta, eps, res, out, (__bf16*)out_hi, (__bf16*)out_lo, rows, dim);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_swin_window_attn(const float* qkv, const float* bias_kzero, const float* scale, const float* rpbT, void* out_hi, void* out_lo,
                                      int N, int H, int W, int C, int heads, int shift, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(qkv && bias_kzero && scale && rpbT && out_hi && N > 0 && H > 0 && W > 0, "swin_window_attn: bad args");
  STEDM_CHECK_ARG(heads > 0 && C == heads * 32, "swin_window_attn: head dim must be 32 (C=%d heads=%d): swin_v2_t/s/b", C, heads);
  STEDM_CHECK_ARG(shift >= 0 && shift < 8, "swin_window_attn: shift %d outside the 8 x 8 window", shift);
  STEDM_CHECK_ARG(N <= 65535 && heads <= 65535, "swin_window_attn: grid limits (N=%d)", N);
  SwinAttnArgs a;
  a.qkv = qkv; a.bias = bias_kzero; a.scale = scale; a.rpbT = rpbT; a.hi = out_hi; a.lo = out_lo;
  a.H = H; a.W = W; a.C = C; a.heads = heads;
  a.padH = (H + 7) / 8 * 8; a.padW = (W + 7) / 8 * 8;
  // "if window size is larger than feature size, there is no need to shift window" (torchvision shifted_window_attention)
  a.shift_h = 8 >= a.padH ? 0 : shift;
  a.shift_w = 8 >= a.padW ? 0 : shift;
  const dim3 grid((a.padH / 8) * (a.padW / 8), heads, N);
  if (mm_dtype == STEDM_F16) swin_window_attn_kernel<_Float16><<<grid, 64, 0, as_stream(stream)>>>(a);
  else swin_window_attn_kernel<__bf16><<<grid, 64, 0, as_stream(stream)>>>(a);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_swin_merge16(const float* x, int N, int H, int W, int C, void* out_hi, void* out_lo, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(x && out_hi && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "swin_merge16: bad args");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)N * Ho * Wo * C;   // 4 parts x C/4 quads per output token
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (mm_dtype == STEDM_F16)
    swin_merge16_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>(x, (_Float16*)out_hi, (_Float16*)out_lo, H, W, C, Ho, Wo, total);
  else
    swin_merge16_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>(x, (__bf16*)out_hi, (__bf16*)out_lo, H, W, C, Ho, Wo, total);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_swin_token_mean(const float* x, float* out, int N, int T, int C, void* stream) {
  STEDM_CHECK_ARG(x && out && N > 0 && T > 0 && C > 0, "swin_token_mean: bad args");
  swin_token_mean_kernel<<<dim3(N, (C + 63) / 64), 256, 0, as_stream(stream)>>>(x, out, T, C);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_swin_rpb(const float* cpb, const long* index, float* rpbT, int heads, int ntab, void* stream) {
  STEDM_CHECK_ARG(cpb && index && rpbT && heads > 0 && ntab > 0, "swin_rpb: bad args");
  swin_rpb_kernel<<<(heads * 4096 + 255) / 256, 256, 0, as_stream(stream)>>>(cpb, index, rpbT, heads, ntab);
  STEDM_LAUNCH_CHECK();
  return 0;
}
