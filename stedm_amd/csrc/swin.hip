// Swin-Transformer-V2 style embedder: the pieces of torchvision's `swin_v2_t` (pinned torchvision==0.18.1,
// call sites networks/s_zss_dm.py:19-20, networks/agg_blocks.py:28,49,70) that are not GEMMs. torchvision is a third-party dependency
// absent from /root/reference: the kernels restate its published algorithm (Liu et al., "Swin Transformer V2", and the layer
// definitions of torchvision.models.swin_transformer) — parity unpinned.
//
//   swin_patch16      : features[0][0] Conv2d(3, 96, 4, stride 4) as a GEMM: 4x4x3 patch gather -> 16-bit operand rows [tok][64]
//                       (k = c*16 + ky*4 + kx, the OIHW order of the conv weight; columns 48..63 zero), any input strides
//   swin_ln           : out = res + LayerNorm(y) (post-norm residual of SwinTransformerBlockV2; res NULL: the plain LayerNorms of
//                       features[0][2], PatchMergingV2.norm and the final norm) as fp32 rows and / or 16-bit operand planes
//   swin_window_attn  : shifted_window_attention with cosine logits (ShiftedWindowAttentionV2) on MFMA, one wave per (window, head): cyclic
//                       shift + window partition as index arithmetic, F.normalize(q) . F.normalize(k) * exp(min(logit_scale, log 100)) +
//                       16 sigmoid(cpb) + shift mask, softmax, P V, written back at the tokens' own positions as the 16-bit plane `proj`
//                       consumes. Windows that hang over the feature map (F.pad) see zero rows: q = q_bias, k = 0, v = v_bias, exactly as
//                       the padded Linear gives.
//   swin_merge16      : PatchMergingV2's 2x2 neighbourhood concat [x(0,0) | x(1,0) | x(0,1) | x(1,1)] -> 16-bit operand rows [tok][4C]
//   swin_token_mean   : AdaptiveAvgPool2d(1) over the tokens of an image
#include <float.h>

#include "conv_common.hpp"
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

// A product that is split into hi / lo planes goes through this first: hipcc otherwise folds `(T)(x * y)` into v_fma_mixlo_f16 (ONE rounding)
// where it rematerialises the hi part and keeps mul + cvt (two roundings) where it stores it; near a tie the stored hi and the hi its lo
// was taken against then differ by an fp16 ulp.
__device__ __forceinline__ float rounded(float v) {
  asm volatile("" : "+v"(v));
  return v;
}

// LPR lanes per row (32: two rows per wave for dim <= 192; 64 above), the row lives in registers (dim <= 12 * LPR): one global read per element
template <typename T, int LPR>
__global__ void __launch_bounds__(256) swin_ln_kernel(const float* __restrict__ y, const float* __restrict__ g, const float* __restrict__ bt,
                                                      float eps, const float* __restrict__ res, float* __restrict__ out, T* __restrict__ hi,
                                                      T* __restrict__ lo, long rows, int dim, int ld16, const float* __restrict__ gate, int rows_per_gate) {
  constexpr int RPB = 256 / LPR, NVMAX = 12;
  const long row = (long)blockIdx.x * RPB + threadIdx.x / LPR;
  if (row >= rows) return;
  const int l = threadIdx.x % LPR;
  const float* py = y + row * dim;
  float v[NVMAX];
  float s1 = 0.f;
#pragma unroll
  for (int j = 0; j < NVMAX; ++j) {
    const int k = l + j * LPR;
    v[j] = k < dim ? py[k] : 0.f;
    s1 += v[j];
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
  const float mean = s1 / dim;
  float s2 = 0.f;
#pragma unroll
  for (int j = 0; j < NVMAX; ++j) {
    const float d = l + j * LPR < dim ? v[j] - mean : 0.f;
    s2 += d * d;
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
  const float rstd = 1.0f / sqrtf(s2 / dim + eps);
  // train-mode stochastic depth (torchvision StochasticDepth, mode "row"): the residual branch of image row / rows_per_gate is scaled by its gate
  const float gt = gate ? gate[row / rows_per_gate] : 1.0f;
#pragma unroll
  for (int j = 0; j < NVMAX; ++j) {
    const int k = l + j * LPR;
    if (k < dim) {
      float w = (v[j] - mean) * rstd * g[k] + bt[k];
      if (gate) w *= gt;
      if (res) w += res[row * dim + k];
      w = rounded(w);
      if (out) out[row * dim + k] = w;
      if (hi) {
        const T h = (T)w;
        hi[row * ld16 + k] = h;
        if (lo) lo[row * ld16 + k] = (T)(w - (float)h);
      }
    }
  }
}

struct SwinAttnArgs {
  const float* qkv;     // [N*H*W][3C]  ('(three heads d)' columns, bias included); NULL with IN16
  const void* qkv16;    // the same rows as 16-bit values of the MFMA type (the qkv GEMM's 16-bit output: half the bytes of the largest stream)
  const float* bias;    // [3C] the Linear's bias with the k third zeroed (pad rows)
  const float* scale;   // [heads] exp(min(logit_scale, log 100))
  const float* rpb;     // [heads][64 queries][64 keys] 16 * sigmoid(cpb_mlp(table))[index]
  void* hi;             // [N*H*W][C]
  void* lo;
  int H, W, C, ld16, heads, shift_h, shift_w, padH, padW, nprob;
};

// One wave per (window, head), 4 per block: the 64 x 64 x 32 attention of an 8 x 8 window on the wave's MFMAs (the form of attn64_mfma_kernel,
// attn.hip). The wave gathers its 64 token rows (coalesced: 8 lanes x 16 B per row), L2-normalises q and k in fp32, rounds to the MFMA type
// (NPASS 3: hi + lo planes, products hi.hi + hi.lo + lo.hi) and keeps q, k [64][32] and V^T [32][64] in LDS.
//   S^T = K Q^T (keys on the rows: a query's softmax is a reduction over the lane's registers plus one cross-half exchange), logits =
//   S^T * scale + bias (+ -100 across shift regions), P = exp(. - max) stays in the accumulators and is the B operand of O^T = V^T P as it
//   lies: k-slot j of k-step (it, u) of lane half h is key (j&3) + 8(2u + (j>>2)) + 4h + 32 it, and V^T is read in that order.
template <typename T, int NPASS, bool IN16 = false>
__global__ void __launch_bounds__(256) swin_window_attn_kernel(const SwinAttnArgs a) {
  using V8 = typename MM<T>::V8;
  typedef T V4T __attribute__((ext_vector_type(4)));
  constexpr int QS = 40, VS = 72;                       // row strides in elements (80 B / 144 B: 16-B resp. 8-B aligned rows, skewed banks)
  constexpr int PLANE = 2 * 64 * QS + 32 * VS;
  constexpr int NPL = NPASS == 3 ? 2 : 1;
  // split-product mode: operands are scaled by exact powers of two before the hi / lo split so that the lo parts (2^-12 of the value) stay
  // above fp16's normal range — normalised q, k entries are ~0.2 and P <= 1, whose lo parts would be subnormal; folded back into the
  // logits scale and the softmax normalisation
  constexpr float QKS = NPASS == 3 ? 64.f : 1.f, PS = NPASS == 3 ? 1024.f : 1.f, VSC = NPASS == 3 ? 64.f : 1.f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* sq = reinterpret_cast<T*>(smem_) + wave * NPL * PLANE;
  T* sk = sq + 64 * QS;
  T* sv = sk + 64 * QS;
  int* sid = reinterpret_cast<int*>(reinterpret_cast<T*>(smem_) + 4 * NPL * PLANE) + wave * 64;
  int prob = blockIdx.x * 4 + wave;
  const bool live = prob < a.nprob;
  if (!live) prob = a.nprob - 1;
  const int nwx = a.padW >> 3, nW = nwx * (a.padH >> 3);
  const int win = prob % nW, hd = (prob / nW) % a.heads;
  const long n = prob / (nW * a.heads);
  const int wy = win / nwx, wx = win % nwx;
  const int C = a.C, c0 = hd * 32;
  const bool masked = (a.shift_h | a.shift_w) != 0;
  // token t of the window (row t >> 3, column t & 7 of the rolled frame) -> source token (torch.roll(x, -shift): rolled[ys] = x[(ys + shift) % pad])
  auto source = [&](int t, long& tok) {
    int y = wy * 8 + (t >> 3) + a.shift_h; if (y >= a.padH) y -= a.padH;
    int x = wx * 8 + (t & 7) + a.shift_w; if (x >= a.padW) x -= a.padW;
    tok = (n * a.H + y) * a.W + x;
    return y < a.H && x < a.W;
  };
  {
    const int sub = lane & 7, tg = lane >> 3;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = 8 * i + tg;
      long tok;
      const bool valid = source(t, tok);
      float4 q4, k4, v4;
      if (IN16 && valid) {
        const T* p16 = reinterpret_cast<const T*>(a.qkv16) + tok * (3L * C) + c0 + 4 * sub;
        const V4T qv = *reinterpret_cast<const V4T*>(p16), kv = *reinterpret_cast<const V4T*>(p16 + C), vv = *reinterpret_cast<const V4T*>(p16 + 2 * C);
        q4 = make_float4((float)qv[0], (float)qv[1], (float)qv[2], (float)qv[3]);
        k4 = make_float4((float)kv[0], (float)kv[1], (float)kv[2], (float)kv[3]);
        v4 = make_float4((float)vv[0], (float)vv[1], (float)vv[2], (float)vv[3]);
      } else {
        const float* pq = ((valid && !IN16) ? a.qkv + tok * (3L * C) : a.bias) + c0 + 4 * sub;      // F.pad rows: the Linear of a zero row is its bias
        q4 = *reinterpret_cast<const float4*>(pq);
        k4 = *reinterpret_cast<const float4*>(pq + C);
        v4 = *reinterpret_cast<const float4*>(pq + 2 * C);
      }
      float nq = q4.x * q4.x + q4.y * q4.y + q4.z * q4.z + q4.w * q4.w;
      float nk = k4.x * k4.x + k4.y * k4.y + k4.z * k4.z + k4.w * k4.w;
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) { nq += __shfl_xor(nq, o, 64); nk += __shfl_xor(nk, o, 64); }
      // F.normalize(p=2, eps=1e-12): v / max(||v||, eps)
      const float rq = QKS / fmaxf(sqrtf(nq), 1e-12f), rk = QKS / fmaxf(sqrtf(nk), 1e-12f);
      float qf[4] = {q4.x * rq, q4.y * rq, q4.z * rq, q4.w * rq};
      float kf[4] = {k4.x * rk, k4.y * rk, k4.z * rk, k4.w * rk};
      float vf[4] = {v4.x * VSC, v4.y * VSC, v4.z * VSC, v4.w * VSC};
      if (NPASS == 3) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { qf[u] = rounded(qf[u]); kf[u] = rounded(kf[u]); vf[u] = rounded(vf[u]); }
      }
      V4T qh, kh, ql, kl;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        qh[u] = (T)qf[u]; kh[u] = (T)kf[u];
        ql[u] = (T)(qf[u] - (float)qh[u]); kl[u] = (T)(kf[u] - (float)kh[u]);
        const T vh = (T)vf[u];
        sv[(4 * sub + u) * VS + t] = vh;
        if (NPASS == 3) sv[PLANE + (4 * sub + u) * VS + t] = (T)(vf[u] - (float)vh);
      }
      *reinterpret_cast<V4T*>(sq + t * QS + 4 * sub) = qh;
      *reinterpret_cast<V4T*>(sk + t * QS + 4 * sub) = kh;
      if (NPASS == 3) {
        *reinterpret_cast<V4T*>(sq + PLANE + t * QS + 4 * sub) = ql;
        *reinterpret_cast<V4T*>(sk + PLANE + t * QS + 4 * sub) = kl;
      }
      if (masked && sub == 0) {
        // shift-mask regions of the rolled frame: [0, pad-8) / [pad-8, pad-shift) / [pad-shift, pad) per side (an unshifted side is one
        // region: torchvision's third slice [-0:] then covers, and overwrites, the whole side)
        const int ys = wy * 8 + i, xs = wx * 8 + tg;
        const int ih = !a.shift_h ? 0 : (ys < a.padH - 8 ? 0 : (ys < a.padH - a.shift_h ? 1 : 2));
        const int iw = !a.shift_w ? 0 : (xs < a.padW - 8 ? 0 : (xs < a.padW - a.shift_w ? 1 : 2));
        sid[t] = ih * 3 + iw;
      }
    }
  }
  __syncthreads();
  const int r = lane & 31, h = lane >> 5;
  f32x16 st[2][2];
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int e = 0; e < 16; ++e) st[it][jt][e] = 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    V8 ka[2], qb[2], kal[2], qbl[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      ka[it] = *reinterpret_cast<const V8*>(sk + (32 * it + r) * QS + 16 * s + 8 * h);
      qb[it] = *reinterpret_cast<const V8*>(sq + (32 * it + r) * QS + 16 * s + 8 * h);
      if (NPASS == 3) {
        kal[it] = *reinterpret_cast<const V8*>(sk + PLANE + (32 * it + r) * QS + 16 * s + 8 * h);
        qbl[it] = *reinterpret_cast<const V8*>(sq + PLANE + (32 * it + r) * QS + 16 * s + 8 * h);
      }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        st[it][jt] = MM<T>::mfma(ka[it], qb[jt], st[it][jt]);
        if (NPASS == 3) {
          st[it][jt] = MM<T>::mfma(ka[it], qbl[jt], st[it][jt]);
          st[it][jt] = MM<T>::mfma(kal[it], qb[jt], st[it][jt]);
        }
      }
  }
  // logits of query column q = 32 jt + r over keys (e&3) + 8(e>>2) + 4h + 32 it; softmax: 32 values here, 32 in lane ^ 32
  const float sc = a.scale[hd] * (1.0f / (QKS * QKS));
  float inv[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int q = 32 * jt + r;
    const float* rp = a.rpb + ((long)hd * 64 + q) * 64 + 4 * h;
    const int rq = masked ? sid[q] : 0;
    float m = -FLT_MAX;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int eq = 0; eq < 4; ++eq) {
        const float4 b4 = *reinterpret_cast<const float4*>(rp + 32 * it + 8 * eq);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v = st[it][jt][4 * eq + k] * sc + bb[k];
          if (masked && sid[32 * it + 8 * eq + 4 * h + k] != rq) v += -100.0f;
          st[it][jt][4 * eq + k] = v;
          m = fmaxf(m, v);
        }
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __expf(st[it][jt][e] - m);
        st[it][jt][e] = pv;
        sum += pv;
      }
    sum += __shfl_xor(sum, 32, 64);
    inv[jt] = 1.0f / (sum * (PS * VSC));
  }
  // O^T[d][q] = sum_k V[k][d] P[k][q]
  f32x16 ot[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int e = 0; e < 16; ++e) ot[jt][e] = 0.f;
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      V8 pb[2], pbl[2];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float pv = NPASS == 3 ? rounded(st[it][jt][8 * u + j] * PS) : st[it][jt][8 * u + j];
          pb[jt][j] = (T)pv;
          if (NPASS == 3) pbl[jt][j] = (T)(pv - (float)pb[jt][j]);
        }
      const T* vrow = sv + r * VS + 32 * it + 16 * u + 4 * h;      // keys (j&3) + 8(2u + (j>>2)) + 4h + 32 it of channel r
      const V4T v0 = *reinterpret_cast<const V4T*>(vrow), v1 = *reinterpret_cast<const V4T*>(vrow + 8);
      V8 va, val;
#pragma unroll
      for (int j = 0; j < 4; ++j) { va[j] = v0[j]; va[4 + j] = v1[j]; }
      if (NPASS == 3) {
        const V4T l0 = *reinterpret_cast<const V4T*>(vrow + PLANE), l1 = *reinterpret_cast<const V4T*>(vrow + PLANE + 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) { val[j] = l0[j]; val[4 + j] = l1[j]; }
      }
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        ot[jt] = MM<T>::mfma(va, pb[jt], ot[jt]);
        if (NPASS == 3) {
          ot[jt] = MM<T>::mfma(va, pbl[jt], ot[jt]);
          ot[jt] = MM<T>::mfma(val, pb[jt], ot[jt]);
        }
      }
    }
  if (!live) return;
  // lane (query 32 jt + r) holds channels (e&3) + 8(e>>2) + 4h: 4 consecutive channels per e-quad -> 8-B stores at the token's own row
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    long tok;
    if (!source(32 * jt + r, tok)) continue;
    T* ph = reinterpret_cast<T*>(a.hi) + tok * a.ld16 + c0 + 4 * h;
    T* pl = a.lo ? reinterpret_cast<T*>(a.lo) + tok * a.ld16 + c0 + 4 * h : nullptr;
#pragma unroll
    for (int eq = 0; eq < 4; ++eq) {
      V4T vh, vl;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v = NPASS == 3 ? rounded(ot[jt][eq * 4 + k] * inv[jt]) : ot[jt][eq * 4 + k] * inv[jt];
        vh[k] = (T)v;
        vl[k] = (T)(v - (float)vh[k]);
      }
      *reinterpret_cast<V4T*>(ph + 8 * eq) = vh;
      if (pl) *reinterpret_cast<V4T*>(pl + 8 * eq) = vl;
    }
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

// rpb[h][query i][key j] = 16 sigmoid(cpb[index[i * 64 + j]][h]) (ShiftedWindowAttentionV2.get_relative_position_bias)
__global__ void __launch_bounds__(256) swin_rpb_kernel(const float* __restrict__ cpb, const long* __restrict__ index, float* __restrict__ out,
                                                       int heads, int ntab) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // over heads * 4096
  if (i >= heads * 4096) return;
  const int h = i >> 12;
  long e = index[i & 4095];
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

static int swin_ln_launch(const float* y, const float* gamma, const float* beta, float eps, const float* res, float* out, void* out_hi,
                          void* out_lo, long rows, int dim, int ld16, const float* gate, int rows_per_gate, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(y && gamma && beta && (out || out_hi) && rows > 0 && dim > 0, "swin_ln: bad args");
  STEDM_CHECK_ARG(!out_lo || out_hi, "swin_ln: out_lo without out_hi");
  STEDM_CHECK_ARG(dim <= 768, "swin_ln: rows of up to 768 channels (dim=%d)", dim);
  STEDM_CHECK_ARG(!out_hi || ld16 >= dim, "swin_ln: ld16 %d < dim %d", ld16, dim);
  hipStream_t st = as_stream(stream);
#define LAUNCH_SWIN_LN(TT)                                                                                                                        \
  {                                                                                                                                               \
    if (dim <= 192) swin_ln_kernel<TT, 32><<<(unsigned)((rows + 7) / 8), 256, 0, st>>>(y, gamma, beta, eps, res, out, (TT*)out_hi, (TT*)out_lo, rows, dim, ld16, gate, rows_per_gate); \
    else swin_ln_kernel<TT, 64><<<(unsigned)((rows + 3) / 4), 256, 0, st>>>(y, gamma, beta, eps, res, out, (TT*)out_hi, (TT*)out_lo, rows, dim, ld16, gate, rows_per_gate);          \
  }
  if (mm_dtype == STEDM_F16) LAUNCH_SWIN_LN(_Float16) else LAUNCH_SWIN_LN(__bf16)
#undef LAUNCH_SWIN_LN
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_swin_ln(const float* y, const float* gamma, const float* beta, float eps, const float* res, float* out, void* out_hi,
                             void* out_lo, long rows, int dim, int ld16, int mm_dtype, void* stream) {
  return swin_ln_launch(y, gamma, beta, eps, res, out, out_hi, out_lo, rows, dim, ld16, nullptr, 1, mm_dtype, stream);
}

extern "C" int stedm_swin_ln_gated(const float* y, const float* gamma, const float* beta, float eps, const float* res, float* out, void* out_hi,
                                   void* out_lo, long rows, int dim, int ld16, const float* gate, int rows_per_gate, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(gate && rows_per_gate > 0 && rows % rows_per_gate == 0, "swin_ln_gated: rows must be a multiple of rows_per_gate");
  return swin_ln_launch(y, gamma, beta, eps, res, out, out_hi, out_lo, rows, dim, ld16, gate, rows_per_gate, mm_dtype, stream);
}

extern "C" int stedm_swin_window_attn(const float* qkv, const void* qkv16, const float* bias_kzero, const float* scale, const float* rpb, void* out_hi, void* out_lo,
                                      int ld16, int N, int H, int W, int C, int heads, int shift, int npass, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG((qkv || qkv16) && bias_kzero && scale && rpb && out_hi && N > 0 && H > 0 && W > 0, "swin_window_attn: bad args");
  STEDM_CHECK_ARG(!qkv16 || (npass == 1 && !qkv), "swin_window_attn: the 16-bit qkv input belongs to the single-product modes (and excludes the fp32 one)");
  STEDM_CHECK_ARG(heads > 0 && C == heads * 32, "swin_window_attn: head dim must be 32 (C=%d heads=%d): swin_v2_t/s/b", C, heads);
  STEDM_CHECK_ARG(shift >= 0 && shift < 8, "swin_window_attn: shift %d outside the 8 x 8 window", shift);
  STEDM_CHECK_ARG(ld16 >= C && ld16 % 4 == 0, "swin_window_attn: ld16 %d (C=%d)", ld16, C);
  STEDM_CHECK_ARG((npass == 1 || npass == 3) && (npass == 1 || out_lo), "swin_window_attn: npass must be 1 or 3 (3 writes out_lo)");
  SwinAttnArgs a;
  a.qkv = qkv; a.qkv16 = qkv16; a.bias = bias_kzero; a.scale = scale; a.rpb = rpb; a.hi = out_hi; a.lo = npass == 3 ? out_lo : nullptr;
  a.H = H; a.W = W; a.C = C; a.ld16 = ld16; a.heads = heads;
  a.padH = (H + 7) / 8 * 8; a.padW = (W + 7) / 8 * 8;
  // "if window size is larger than feature size, there is no need to shift window" (torchvision shifted_window_attention)
  a.shift_h = 8 >= a.padH ? 0 : shift;
  a.shift_w = 8 >= a.padW ? 0 : shift;
  const long nprob = (long)N * (a.padH / 8) * (a.padW / 8) * heads;
  STEDM_CHECK_ARG(nprob < (1L << 31), "swin_window_attn: too many windows (%ld)", nprob);
  a.nprob = (int)nprob;
  const unsigned grid = (unsigned)((nprob + 3) / 4);
  constexpr int kPlaneBytes = (2 * 64 * 40 + 32 * 72) * 2;
  const size_t lds = (size_t)4 * (npass == 3 ? 2 : 1) * kPlaneBytes + 4 * 64 * sizeof(int);
  hipStream_t st = as_stream(stream);
#define LAUNCH_SWIN_ATTN(TT, NP)                                                                                                          \
  {                                                                                                                                       \
    if (lds > 64 * 1024)                                                                                                                  \
      STEDM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(swin_window_attn_kernel<TT, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    swin_window_attn_kernel<TT, NP><<<grid, 256, lds, st>>>(a);                                                                           \
  }
  if (qkv16) {
    if (mm_dtype == STEDM_F16) swin_window_attn_kernel<_Float16, 1, true><<<grid, 256, lds, st>>>(a);
    else swin_window_attn_kernel<__bf16, 1, true><<<grid, 256, lds, st>>>(a);
  } else if (mm_dtype == STEDM_F16) { if (npass == 3) LAUNCH_SWIN_ATTN(_Float16, 3) else LAUNCH_SWIN_ATTN(_Float16, 1) }
  else { if (npass == 3) LAUNCH_SWIN_ATTN(__bf16, 3) else LAUNCH_SWIN_ATTN(__bf16, 1) }
#undef LAUNCH_SWIN_ATTN
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

extern "C" int stedm_swin_rpb(const float* cpb, const long* index, float* rpb, int heads, int ntab, void* stream) {
  STEDM_CHECK_ARG(cpb && index && rpb && heads > 0 && ntab > 0, "swin_rpb: bad args");
  swin_rpb_kernel<<<(heads * 4096 + 255) / 256, 256, 0, as_stream(stream)>>>(cpb, index, rpb, heads, ntab);
  STEDM_LAUNCH_CHECK();
  return 0;
}
