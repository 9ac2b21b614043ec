// QKVAttentionLegacy (openaimodel.py:378-394) as a flash-style fp32 kernel: never materialises
// the [T][T] weights. qkv is [B][T][heads*3*ch] (token-major, the NHWC view of the reference's
// [B][3C][T]) with channel = h*3*ch + {q:0, k:ch, v:2*ch} + c. Softmax runs in fp32 exactly as the
// reference's `softmax(weight.float())`; q and k are each scaled by ch^-1/4 before the product.
//
// v1: VALU fp32 (the U-Net's middle attention is 0.03 % of the forward's FLOPs at 32x32 latents);
// one 256-thread block per (sample*head, 64-query tile); K/V tiles of 64 keys streamed through LDS.
#include <stdlib.h>

#include "common.hpp"
using namespace stedm;

constexpr int AT = 64;  // query rows per block == keys per tile

__global__ void __launch_bounds__(256) attn_legacy_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T,
                                                          int heads, int ch, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smf[];
  const int LD = ch + 4;       // row stride (floats), keeps 16-B alignment
  float* Qs = smf;             // [AT][LD]
  float* Ks = Qs + AT * LD;    // [AT][LD]
  float* Vs = Ks + AT * LD;    // [AT][LD]
  float* Ps = Vs + AT * LD;    // [AT][AT+4]
  constexpr int PLD = AT + 4;

  const int bh = blockIdx.x, b = bh / heads, hd = bh % heads;
  const int q0 = blockIdx.y * AT;
  const int C3 = heads * 3 * ch, C = heads * ch;
  const float* base = qkv + (long)b * T * C3 + hd * 3 * ch;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int c4n = ch >> 2;

  for (int i = tid; i < AT * c4n; i += 256) {
    const int row = i / c4n, c4 = i - row * c4n;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q0 + row < T) {
      v = *reinterpret_cast<const float4*>(base + (long)(q0 + row) * C3 + c4 * 4);
      v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    }
    *reinterpret_cast<float4*>(Qs + row * LD + c4 * 4) = v;
  }

  float m_run[4], l_run[4];
  float4 o[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    m_run[i] = -INFINITY;
    l_run[i] = 0.f;
    o[i][0] = make_float4(0.f, 0.f, 0.f, 0.f);
    o[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  for (int k0 = 0; k0 < T; k0 += AT) {
    __syncthreads();  // previous tile fully consumed (also covers the Q stores on the first pass)
    for (int i = tid; i < AT * c4n; i += 256) {
      const int row = i / c4n, c4 = i - row * c4n;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (k0 + row < T) {
        const float* pr = base + (long)(k0 + row) * C3 + c4 * 4;
        kv = *reinterpret_cast<const float4*>(pr + ch);
        vv = *reinterpret_cast<const float4*>(pr + 2 * ch);
        kv.x *= scale; kv.y *= scale; kv.z *= scale; kv.w *= scale;
      }
      *reinterpret_cast<float4*>(Ks + row * LD + c4 * 4) = kv;
      *reinterpret_cast<float4*>(Vs + row * LD + c4 * 4) = vv;
    }
    __syncthreads();

    // S = Q K^T : thread owns rows ty*4+i, cols tx*4+j
    float s[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s[i][j] = 0.f;
    for (int c = 0; c < ch; c += 4) {
      float4 qv[4], kv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) qv[i] = *reinterpret_cast<const float4*>(Qs + (ty * 4 + i) * LD + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) kv[j] = *reinterpret_cast<const float4*>(Ks + (tx * 4 + j) * LD + c);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[i][j] = fmaf(qv[i].x, kv[j].x, s[i][j]);
          s[i][j] = fmaf(qv[i].y, kv[j].y, s[i][j]);
          s[i][j] = fmaf(qv[i].z, kv[j].z, s[i][j]);
          s[i][j] = fmaf(qv[i].w, kv[j].w, s[i][j]);
        }
    }
    // online softmax (row statistics shared by the 16 lanes with equal ty)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (k0 + tx * 4 + j >= T) s[i][j] = -INFINITY;
        mx = fmaxf(mx, s[i][j]);
      }
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
      const float m_new = fmaxf(m_run[i], mx);
      const float alpha = __expf(m_run[i] - m_new);  // exp(-inf) = 0 on the first tile
      float rs = 0.f;
      float4 pv;
      pv.x = __expf(s[i][0] - m_new); pv.y = __expf(s[i][1] - m_new);
      pv.z = __expf(s[i][2] - m_new); pv.w = __expf(s[i][3] - m_new);
      rs = (pv.x + pv.y) + (pv.z + pv.w);
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) rs += __shfl_xor(rs, d, 64);
      l_run[i] = l_run[i] * alpha + rs;
      m_run[i] = m_new;
      o[i][0].x *= alpha; o[i][0].y *= alpha; o[i][0].z *= alpha; o[i][0].w *= alpha;
      o[i][1].x *= alpha; o[i][1].y *= alpha; o[i][1].z *= alpha; o[i][1].w *= alpha;
      *reinterpret_cast<float4*>(Ps + (ty * 4 + i) * PLD + tx * 4) = pv;
    }
    __syncthreads();
    // O += P V : thread owns rows ty*4+i, channel quads tx + 16*jj
    for (int sidx = 0; sidx < AT; sidx += 4) {
      float4 pr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pr[i] = *reinterpret_cast<const float4*>(Ps + (ty * 4 + i) * PLD + sidx);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int c4 = tx + 16 * jj;
        if (c4 < c4n) {
          const float4 v0 = *reinterpret_cast<const float4*>(Vs + (sidx + 0) * LD + c4 * 4);
          const float4 v1 = *reinterpret_cast<const float4*>(Vs + (sidx + 1) * LD + c4 * 4);
          const float4 v2 = *reinterpret_cast<const float4*>(Vs + (sidx + 2) * LD + c4 * 4);
          const float4 v3 = *reinterpret_cast<const float4*>(Vs + (sidx + 3) * LD + c4 * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float4& a = o[i][jj];
            a.x = fmaf(pr[i].x, v0.x, a.x); a.y = fmaf(pr[i].x, v0.y, a.y); a.z = fmaf(pr[i].x, v0.z, a.z); a.w = fmaf(pr[i].x, v0.w, a.w);
            a.x = fmaf(pr[i].y, v1.x, a.x); a.y = fmaf(pr[i].y, v1.y, a.y); a.z = fmaf(pr[i].y, v1.z, a.z); a.w = fmaf(pr[i].y, v1.w, a.w);
            a.x = fmaf(pr[i].z, v2.x, a.x); a.y = fmaf(pr[i].z, v2.y, a.y); a.z = fmaf(pr[i].z, v2.z, a.z); a.w = fmaf(pr[i].z, v2.w, a.w);
            a.x = fmaf(pr[i].w, v3.x, a.x); a.y = fmaf(pr[i].w, v3.y, a.y); a.z = fmaf(pr[i].w, v3.z, a.z); a.w = fmaf(pr[i].w, v3.w, a.w);
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = q0 + ty * 4 + i;
    if (t >= T) continue;
    const float inv = 1.0f / l_run[i];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int c4 = tx + 16 * jj;
      if (c4 < c4n) {
        float4 v = o[i][jj];
        v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
        *reinterpret_cast<float4*>(out + ((long)b * T + t) * C + hd * ch + c4 * 4) = v;
      }
    }
  }
}

extern "C" int stedm_attn_legacy(const float* qkv, float* out, int B, int T, int heads, int ch, void* stream) {
  STEDM_CHECK_ARG(qkv && out, "attn_legacy: null pointer");
  STEDM_CHECK_ARG(B > 0 && T > 0 && heads > 0, "attn_legacy: bad sizes");
  STEDM_CHECK_ARG(ch % 4 == 0 && ch >= 4 && ch <= 128, "attn_legacy: head width %d unsupported (need ch %% 4 == 0, ch <= 128)", ch);
  const size_t lds = ((size_t)3 * AT * (ch + 4) + (size_t)AT * (AT + 4)) * sizeof(float);
  if (lds > 64 * 1024)
    STEDM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_legacy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const float scale = 1.0f / sqrtf(sqrtf((float)ch));
  dim3 grid(B * heads, (T + AT - 1) / AT);
  attn_legacy_kernel<<<grid, 256, lds, as_stream(stream)>>>(qkv, out, T, heads, ch, scale);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// T = 64 tokens (the 8x8 middle block), single-product modes: the whole attention of one (sample, head) on one wave's MFMAs.
//   S^T = K Q^T (keys on the rows: the softmax over keys of a query is a reduction over the lane's own registers plus one
//   cross-half exchange), P = exp(scale^2 S^T - max) stays in the accumulators and is the B operand of O^T = V^T P as it lies:
//   k-slot j of k-step (it, u) of lane half h is key (j&3) + 8(2u + (j>>2)) + 4h + 32 it, and V^T is gathered in that order.
// Operands are rounded to the 16-bit MFMA type (as every other contraction of these modes), logits / softmax / normalisation fp32.
// Output: the 16-bit operand plane [B][64][heads*ch] of proj_out (no fp32 round trip, no conversion pass).
// ------------------------------------------------------------------------------------------------
#include "conv_common.hpp"

// IN16: qkv arrives as the 16-bit plane the qkv convolution's epilogue wrote (same rounding as converting here, half the bytes).
template <typename T, int CH, bool IN16>
__global__ void __launch_bounds__(256) attn64_mfma_kernel(const void* __restrict__ qkv_, T* __restrict__ out, int nprob, int heads, float scale2) {
  using V8 = typename MM<T>::V8;
  typedef T V4T __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int prob = blockIdx.x * 4 + wave;
  if (prob >= nprob) return;
  const int b = prob / heads, hd = prob % heads;
  const int C3 = heads * 3 * CH, C = heads * CH;
  const long boff = (long)b * 64 * C3 + hd * 3 * CH;
  const float* base = reinterpret_cast<const float*>(qkv_) + boff;
  const T* base16 = reinterpret_cast<const T*>(qkv_) + boff;
  const int r = lane & 31, h = lane >> 5;

  auto frag8 = [&](long off) {   // 8 consecutive elements at element offset `off` -> one MFMA fragment
    if (IN16) return *reinterpret_cast<const V8*>(base16 + off);
    const float4 x = *reinterpret_cast<const float4*>(base + off), y = *reinterpret_cast<const float4*>(base + off + 4);
    V8 f;
    f[0] = (T)x.x; f[1] = (T)x.y; f[2] = (T)x.z; f[3] = (T)x.w; f[4] = (T)y.x; f[5] = (T)y.y; f[6] = (T)y.z; f[7] = (T)y.w;
    return f;
  };

  f32x16 st[2][2];
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int e = 0; e < 16; ++e) st[it][jt][e] = 0.f;
#pragma unroll
  for (int s = 0; s < CH / 16; ++s) {
    V8 ka[2], qb[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) ka[it] = frag8((long)(32 * it + r) * C3 + CH + 16 * s + 8 * h);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) qb[jt] = frag8((long)(32 * jt + r) * C3 + 16 * s + 8 * h);
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) st[it][jt] = MM<T>::mfma(ka[it], qb[jt], st[it][jt]);
  }
  // softmax over the keys of query column 32 jt + r: 32 values in this lane, 32 in lane ^ 32
  float inv[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    float m = -INFINITY;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int e = 0; e < 16; ++e) m = fmaxf(m, st[it][jt][e]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __expf((st[it][jt][e] - m) * scale2);
        st[it][jt][e] = pv;
        sum += pv;
      }
    sum += __shfl_xor(sum, 32, 64);
    inv[jt] = 1.0f / sum;
  }
  // O^T[d][q] = sum_k V[k][d] P[k][q]
  f32x16 ot[CH / 32][2];
#pragma unroll
  for (int dt = 0; dt < CH / 32; ++dt)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int e = 0; e < 16; ++e) ot[dt][jt][e] = 0.f;
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      V8 pb[2];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int j = 0; j < 8; ++j) pb[jt][j] = (T)st[it][jt][8 * u + j];
#pragma unroll
      for (int dt = 0; dt < CH / 32; ++dt) {
        V8 va;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int key = (j & 3) + 8 * (2 * u + (j >> 2)) + 4 * h + 32 * it;
          const long vo = (long)key * C3 + 2 * CH + 32 * dt + r;
          va[j] = IN16 ? base16[vo] : (T)base[vo];
        }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) ot[dt][jt] = MM<T>::mfma(va, pb[jt], ot[dt][jt]);
      }
    }
  // lane (column q = 32 jt + r) holds channels d = 32 dt + (e&3) + 8(e>>2) + 4h: 4 consecutive channels per e-quad -> 8-B stores
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    T* orow = out + ((long)b * 64 + 32 * jt + r) * C + hd * CH;
#pragma unroll
    for (int dt = 0; dt < CH / 32; ++dt)
#pragma unroll
      for (int eq = 0; eq < 4; ++eq) {
        V4T v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (T)(ot[dt][jt][eq * 4 + k] * inv[jt]);
        *reinterpret_cast<V4T*>(orow + 32 * dt + 8 * eq + 4 * h) = v;
      }
  }
}

// T = 64 n tokens (the middle block at 64x64 latents: 256; at 128x128: 1024), single-product modes, 16-bit qkv plane in: one wave per
// (sample, head, 32-query block), key tiles of 64 with the online softmax of a flash kernel around the products of attn64_mfma_kernel above
// (running maximum / sum per query in fp32, O^T rescaled only when a maximum moved). 32 queries per wave: with 64 the CH = 128 form needed
// 512 registers and scratch. Replaces the fp32 VALU kernel (1.23 ms per CFG pass of 64 latents of 64x64) for these sizes.
template <typename T, int CH>
__global__ void __launch_bounds__(256) attn_mfma_tiles_kernel(const T* __restrict__ qkv16, T* __restrict__ out, int nprob, int heads, int Tn, float scale2) {
  using V8 = typename MM<T>::V8;
  typedef T V4T __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int prob = blockIdx.x * 4 + wave;
  if (prob >= nprob) return;
  const int nqb = Tn >> 5, nkt = Tn >> 6;
  const int qb = prob % nqb, bh = prob / nqb;
  const int b = bh / heads, hd = bh % heads;
  const int C3 = heads * 3 * CH, C = heads * CH;
  const T* base16 = qkv16 + (long)b * Tn * C3 + hd * 3 * CH;
  const int r = lane & 31, h = lane >> 5;
  auto frag8 = [&](long off) { return *reinterpret_cast<const V8*>(base16 + off); };

  V8 qf[CH / 16];      // the wave's 32 queries, all k-steps
#pragma unroll
  for (int s = 0; s < CH / 16; ++s) qf[s] = frag8((long)(qb * 32 + r) * C3 + 16 * s + 8 * h);
  f32x16 ot[CH / 32];
#pragma unroll
  for (int dt = 0; dt < CH / 32; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) ot[dt][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
#pragma unroll 1
  for (int kt = 0; kt < nkt; ++kt) {
    f32x16 st[2];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int e = 0; e < 16; ++e) st[it][e] = 0.f;
#pragma unroll
    for (int s = 0; s < CH / 16; ++s) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const V8 ka = frag8((long)(kt * 64 + 32 * it + r) * C3 + CH + 16 * s + 8 * h);
        st[it] = MM<T>::mfma(ka, qf[s], st[it]);
      }
    }
    // online softmax over the keys of query column r: 32 values of this tile in this lane, 32 in lane ^ 32
    float m = -INFINITY;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int e = 0; e < 16; ++e) m = fmaxf(m, st[it][e]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float m_new = fmaxf(m_run, m);
    const float alpha = __expf((m_run - m_new) * scale2);
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __expf((st[it][e] - m_new) * scale2);
        st[it][e] = pv;
        sum += pv;
      }
    sum += __shfl_xor(sum, 32, 64);
    l_run = l_run * alpha + sum;
    m_run = m_new;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
      for (int dt = 0; dt < CH / 32; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) ot[dt][e] *= alpha;
    }
    // O^T[d][q] += sum_k V[k][d] P[k][q] over this tile's keys
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        V8 pb;
#pragma unroll
        for (int j = 0; j < 8; ++j) pb[j] = (T)st[it][8 * u + j];
#pragma unroll
        for (int dt = 0; dt < CH / 32; ++dt) {
          V8 va;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int key = kt * 64 + (j & 3) + 8 * (2 * u + (j >> 2)) + 4 * h + 32 * it;
            va[j] = base16[(long)key * C3 + 2 * CH + 32 * dt + r];
          }
          ot[dt] = MM<T>::mfma(va, pb, ot[dt]);
        }
      }
  }
  // lane (query r) holds channels d = 32 dt + (e&3) + 8(e>>2) + 4h: 4 consecutive channels per e-quad -> 8-B stores
  const float inv = 1.0f / l_run;
  T* orow = out + ((long)b * Tn + qb * 32 + r) * C + hd * CH;
#pragma unroll
  for (int dt = 0; dt < CH / 32; ++dt)
#pragma unroll
    for (int eq = 0; eq < 4; ++eq) {
      V4T v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (T)(ot[dt][eq * 4 + k] * inv);
      *reinterpret_cast<V4T*>(orow + 32 * dt + 8 * eq + 4 * h) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Flash form of the same attention for any token count (the middle block at 128 x 128 latents: T = 1024, ch = 128; the SpatialTransformer's
// CrossAttention, ldm/modules/attention.py:170-193, whose stacked to_q | to_k | to_v rows are packed in this head-major order): a workgroup
// owns NW x 32 queries of one (sample, head) and streams the K and V rows of 64-key tiles ONCE through LDS for all its waves
// (global_load_lds_dwordx4 straight from the token-major 16-bit qkv plane, ring of two 32-KB stages, one block barrier per tile) - round 4's
// attn_mfma_tiles_kernel had every wave fetch its own K fragments from global memory and gather V with 2-byte loads (210 TFLOP/s, 8.4 % of
// the MFMA peak at T = 1024). LDS image of a tile: [64 rows][256-B pitch], 16-B chunk ch of row r at chunk ch ^ (((r & 3) << 2) | ((r >> 2) & 3))
// (cdna_hip_programming.md T10 image (b)): the K fragments are row reads (ds_read_b128), the V^T fragments of O^T += V^T P^T are
// transposing reads (ds_read_b64_tr_b16) of the SAME token-major rows, both conflict-free - no V^T copy of the qkv plane exists anywhere.
// S^T = K Q^T is computed swapped (keys on the accumulator rows, the wave's 32 queries on the lanes): the softmax statistics of a query are
// lane-local plus one cross-half exchange, and the exponentiated tile is the B operand of the second product as it lies
// (k-slot j of k-step (it, u) of lane half h is key 32 it + 16 u + 8 (j >> 2) + 4 h + (j & 3)).
// Head widths 16 (zero-extended to one 32-wide MFMA k-step / d-tile), 32, 64, 128; keys beyond T are masked, query rows beyond T are not stored.
// ------------------------------------------------------------------------------------------------
#define GLDS16(gptr, lptr)                                                                                  \
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr),                   \
                                   (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

static __device__ __attribute__((aligned(256))) unsigned char g_attn_zero[256];   // source of the zero extension (ch = 16)

struct AttnFlashArgs {
  const void* qkv;   // 16-bit plane [B][T][heads * 3 * ch], channel = head * 3 ch + {q: 0, k: ch, v: 2 ch} + c
  void* out;         // 16-bit plane [B][T][heads * ch]
  int T, heads, nbh, nq;
  float c;           // logit scale x log2(e): p = exp2((s - m) c)
};

template <typename T, int CH, int NW>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) attn_flash_kernel(const AttnFlashArgs a) {
  using V8 = typename MM<T>::V8;
  typedef T V4T __attribute__((ext_vector_type(4)));
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4* lp4;
  constexpr int DP = CH < 32 ? 32 : CH;          // head width as the MFMAs see it
  constexpr int KS = DP / 16, DT = DP / 32;
  constexpr int NBUF = 2, TILE_B = 32768;        // K rows 16 KiB | V rows 16 KiB
  __shared__ __attribute__((aligned(1024))) unsigned char ring[NBUF * TILE_B];
  // block -> (sample-head, query tile): all query tiles of a sample-head on one XCD (its K / V rows stay in that XCD's L2)
  int bh, qtile;
  {
    const int L = blockIdx.x;
    if ((a.nbh & 7) == 0) { const int x = L & 7, j = L >> 3; bh = x + 8 * (j / a.nq); qtile = j % a.nq; }
    else { bh = L % a.nbh; qtile = L / a.nbh; }
  }
  const int b = bh / a.heads, hd = bh % a.heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int C3 = a.heads * 3 * CH, C = a.heads * CH;
  const T* base = reinterpret_cast<const T*>(a.qkv) + (long)b * a.T * C3 + hd * 3 * CH;
  const int q0 = qtile * (NW * 32) + wave * 32;
  const bool live = q0 < a.T;                    // wave-uniform

  V8 qf[KS];
  {
    const int row = min(q0 + r, a.T - 1);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (16 * s + 8 * h < CH) qf[s] = *reinterpret_cast<const V8*>(base + (long)row * C3 + 16 * s + 8 * h);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[s][j] = (T)0.f;
      }
    }
  }
  // DMA: instruction j of a tile half fills rows 4 j .. 4 j + 3 (1 KiB): lane -> row 4 j + (lane >> 4), physical chunk lane & 15
  constexpr int NI = 16 / NW;                    // instructions per wave, tile and operand
  const int drow = 4 * wave + (lane >> 4);       // + 4 NW i
  auto issue = [&](int kt) __attribute__((always_inline)) {
    unsigned char* dst = ring + (kt % NBUF) * TILE_B + wave * 1024;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = drow + 4 * NW * i;
      const int ch = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));      // logical chunk held by this lane's slot
      const T* src = base + (long)min(kt * 64 + row, a.T - 1) * C3 + ch * 8;
      if (ch * 8 < CH) {
        GLDS16(src + CH, dst + i * (NW * 1024));
        GLDS16(src + 2 * CH, dst + 16384 + i * (NW * 1024));
      } else if (ch * 8 < DP) {
        GLDS16(g_attn_zero + (lane & 15) * 16, dst + i * (NW * 1024));      // zero extension of K (V's extension rows feed output channels that are never stored)
      }
    }
  };
  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int ntiles = (a.T + 63) >> 6;
  // fragment addresses inside a tile (bytes)
  const int swz = ((r & 3) << 2) | ((r >> 2) & 3);
  const unsigned kb = 256u * r + 16u * (unsigned)(h ^ swz);                      // K row read of k-step s: kb ^ (s << 5), + 8192 for rows 32 ..
  const int g1 = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;          // transposed read: lane 4 q + p of its 16-lane group
  const unsigned vlo = 16384u + 256u * (4 * h + tq) + 16u * (unsigned)((tq << 2) | (g1 << 1) | ((tp >> 1) ^ h)) + 8u * (tp & 1);
  const unsigned vhi = 16384u + 256u * (4 * h + tq + 8) + 16u * (unsigned)((tq << 2) | ((g1 ^ 1) << 1) | ((tp >> 1) ^ h)) + 8u * (tp & 1);

  issue(0);
#pragma unroll 1
  for (int kt = 0; kt < ntiles; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();        // tile kt is complete in LDS; every wave is done with tile kt - 1, whose stage takes tile kt + 1
    if (kt + 1 < ntiles) issue(kt + 1);
    if (!live) continue;                 // a wave without queries only feeds the ring
    const unsigned char* tb = ring + (kt % NBUF) * TILE_B;
    f32x16 st[2];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int e = 0; e < 16; ++e) st[it][e] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const V8 ka = *reinterpret_cast<const V8*>(tb + it * 8192 + (kb ^ (unsigned)(s << 5)));
        st[it] = MM<T>::mfma(ka, qf[s], st[it]);
      }
    if (kt * 64 + 64 > a.T) {
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (kt * 64 + 32 * it + (e & 3) + 8 * (e >> 2) + 4 * h >= a.T) st[it][e] = -INFINITY;
    }
    // online softmax over the keys of query column r: 32 values of this tile in this lane, 32 in lane ^ 32
    float m = st[0][0];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int e = 0; e < 16; ++e) m = fmaxf(m, st[it][e]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float m_new = fmaxf(m_run, m);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * a.c);
    const float mc = m_new * a.c;
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(st[it][e], a.c, -mc));
        st[it][e] = pv;
        sum += pv;
      }
    sum += __shfl_xor(sum, 32, 64);
    l_run = l_run * alpha + sum;
    m_run = m_new;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
    }
    // O^T[d][q] += sum_k V[k][d] P[k][q] over this tile's keys
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        V8 pb;
#pragma unroll
        for (int j = 0; j < 8; ++j) pb[j] = (T)st[it][8 * u + j];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const unsigned tofs = 4096u * (2 * it + u);
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(tb + tofs + (vlo ^ (unsigned)(dt << 6))));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(tb + tofs + (vhi ^ (unsigned)(dt << 6))));
          const s16x8 v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          o[dt] = MM<T>::mfma(__builtin_bit_cast(V8, v8), pb, o[dt]);
        }
      }
  }
  // lane (query r) holds channels d = 32 dt + (e & 3) + 8 (e >> 2) + 4 h: 4 consecutive channels per e-quad -> 8-B stores
  if (!live || q0 + r >= a.T) return;
  const float inv = 1.0f / l_run;
  T* orow = reinterpret_cast<T*>(a.out) + ((long)b * a.T + q0 + r) * C + hd * CH;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int eq = 0; eq < 4; ++eq) {
      if (32 * dt + 8 * eq + 4 * h >= CH) continue;
      V4T v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (T)(o[dt][eq * 4 + k] * inv);
      *reinterpret_cast<V4T*>(orow + 32 * dt + 8 * eq + 4 * h) = v;
    }
}

template <typename T, int NW>
static int attn_flash_launch(const AttnFlashArgs& a, int ch, hipStream_t st) {
  const int grid = a.nbh * a.nq;
  if (ch == 128) attn_flash_kernel<T, 128, NW><<<grid, NW * 64, 0, st>>>(a);
  else if (ch == 64) attn_flash_kernel<T, 64, NW><<<grid, NW * 64, 0, st>>>(a);
  else if (ch == 32) attn_flash_kernel<T, 32, NW><<<grid, NW * 64, 0, st>>>(a);
  else attn_flash_kernel<T, 16, NW><<<grid, NW * 64, 0, st>>>(a);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_attn_legacy16(const void* qkv, int qkv_is16, void* out16, int B, int T, int heads, int ch, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(qkv && out16 && B > 0 && heads > 0, "attn_legacy16: bad args");
  STEDM_CHECK_ARG(T > 0 && (ch == 128 || ch == 64 || ch == 32 || ch == 16) && ((T == 64 && ch != 16) || qkv_is16),
                  "attn_legacy16: covers head widths 16 / 32 / 64 / 128 from the 16-bit qkv plane (any T) and T == 64 (ch >= 32) from fp32 rows (T=%d ch=%d is16=%d)", T, ch, qkv_is16);
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "attn_legacy16: bad mm_dtype");
  if (T != 64 || ch == 16) {
    // flash form: workgroups of 4 waves x 32 queries (two per CU)
    hipStream_t s_ = as_stream(stream);
    static const bool tiles_ab = getenv("STEDM_ATTN_TILES") != nullptr;      // round-4 kernel, kept for the A/B of profiles/r05_attn_flash.md
    if (tiles_ab && T % 64 == 0 && ch >= 64) {
      const int np = B * heads * (T / 32), g = (np + 3) / 4;
      const float sc2 = 1.0f / sqrtf((float)ch);
      if (mm_dtype == STEDM_F16) {
        if (ch == 128) attn_mfma_tiles_kernel<_Float16, 128><<<g, 256, 0, s_>>>((const _Float16*)qkv, (_Float16*)out16, np, heads, T, sc2);
        else attn_mfma_tiles_kernel<_Float16, 64><<<g, 256, 0, s_>>>((const _Float16*)qkv, (_Float16*)out16, np, heads, T, sc2);
      } else {
        if (ch == 128) attn_mfma_tiles_kernel<__bf16, 128><<<g, 256, 0, s_>>>((const __bf16*)qkv, (__bf16*)out16, np, heads, T, sc2);
        else attn_mfma_tiles_kernel<__bf16, 64><<<g, 256, 0, s_>>>((const __bf16*)qkv, (__bf16*)out16, np, heads, T, sc2);
      }
      STEDM_LAUNCH_CHECK();
      return 0;
    }
    AttnFlashArgs fa{qkv, out16, T, heads, B * heads, (T + 127) / 128, 1.4426950408889634f / sqrtf((float)ch)};
    return mm_dtype == STEDM_F16 ? attn_flash_launch<_Float16, 4>(fa, ch, s_) : attn_flash_launch<__bf16, 4>(fa, ch, s_);
  }
  const int nprob = B * heads, grid = (nprob + 3) / 4;
  const float scale2 = 1.0f / sqrtf((float)ch);
  hipStream_t st = as_stream(stream);
#define LAUNCH_ATTN(TT, CHH)                                                                                         \
  {                                                                                                                 \
    if (qkv_is16) attn64_mfma_kernel<TT, CHH, true><<<grid, 256, 0, st>>>(qkv, (TT*)out16, nprob, heads, scale2);    \
    else attn64_mfma_kernel<TT, CHH, false><<<grid, 256, 0, st>>>(qkv, (TT*)out16, nprob, heads, scale2);           \
  }
  if (mm_dtype == STEDM_F16) {
    if (ch == 128) LAUNCH_ATTN(_Float16, 128) else if (ch == 64) LAUNCH_ATTN(_Float16, 64) else LAUNCH_ATTN(_Float16, 32)
  } else {
    if (ch == 128) LAUNCH_ATTN(__bf16, 128) else if (ch == 64) LAUNCH_ATTN(__bf16, 64) else LAUNCH_ATTN(__bf16, 32)
  }
#undef LAUNCH_ATTN
  STEDM_LAUNCH_CHECK();
  return 0;
}
