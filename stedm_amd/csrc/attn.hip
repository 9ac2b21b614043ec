// QKVAttentionLegacy (openaimodel.py:378-394) as a flash-style fp32 kernel: never materialises
// the [T][T] weights. qkv is [B][T][heads*3*ch] (token-major, the NHWC view of the reference's
// [B][3C][T]) with channel = h*3*ch + {q:0, k:ch, v:2*ch} + c. Softmax runs in fp32 exactly as the
// reference's `softmax(weight.float())`; q and k are each scaled by ch^-1/4 before the product.
//
// v1: VALU fp32 (the U-Net's middle attention is 0.03 % of the forward's FLOPs at 32x32 latents);
// one 256-thread block per (sample*head, 64-query tile); K/V tiles of 64 keys streamed through LDS.
#include <stdlib.h>

#include <type_traits>

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

// ------------------------------------------------------------------------------------------------
// Flash form of the same attention for any token count (the middle block at 128 x 128 latents: T = 1024, ch = 128; the SpatialTransformer's
// CrossAttention, ldm/modules/attention.py:170-193, whose stacked to_q | to_k | to_v rows are packed in this head-major order): a workgroup
// owns NW x 32 queries of one (sample, head) and streams the K and V rows of 64-key tiles ONCE through LDS for all its waves
// (global_load_lds_dwordx4 straight from the token-major 16-bit qkv plane, rings of 16-KB stages, one block barrier per tile) - round 4's
// attn_mfma_tiles_kernel had every wave fetch its own K fragments from global memory and gather V with 2-byte loads (210 - 240 TFLOP/s,
// 8.4 - 9.7 % of the MFMA peak at T = 1024; this kernel: 750 - 760, profiles/r05_attn_flash.md). The K fragments are row reads
// (ds_read_b128), the V^T fragments of O^T += V^T P^T are transposing reads (ds_read_b64_tr_b16) of the SAME token-major rows, both
// conflict-free on the subtile image described in the kernel (SQ_LDS_BANK_CONFLICT = 0) - no V^T copy of the qkv plane exists anywhere.
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

// Diagnostic builds only (-DSTEDM_ATTN_DIAG=<bits>, tools/attn_diag.sh: extra libraries, never the shipped one): COMPILE-TIME removal of one
// ingredient of attn_flash_kernel's loop at a time - 1: no exponentials (vector work), 2: no fragment reads from LDS, 4: no DMA (tiles are
// never loaded), 8: no block barrier, 16: no MFMAs - to attribute the loop's time. Results are garbage; the shipped build compiles none of it.
#ifdef STEDM_ATTN_DIAG
#define AT_DIAG(BIT) ((STEDM_ATTN_DIAG) & (BIT))
#else
#define AT_DIAG(BIT) 0
#endif

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <typename T, int CH, int NW, int NBUF>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) attn_flash_kernel(const AttnFlashArgs a) {
  using V8 = typename MM<T>::V8;
  typedef T V4T __attribute__((ext_vector_type(4)));
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int DP = CH < 32 ? 32 : CH;          // head width as the MFMAs see it
  constexpr int KS = DP / 16, DT = DP / 32;
  constexpr int NQK = 2 * KS, NPV = 4 * DT;      // MFMAs of a tile's two products (+ 2 reference steps in front of the first)
  constexpr int STAGE = 16384;                   // 64 rows x 256 B as 8 row blocks x 4 chunk groups of 512-B subtiles
  // two rings of NBUF stages: iteration kt multiplies K(kt + 1) and V(kt), so the K and V rows of a tile live on different schedules.
  // NW = 4, NBUF = 2 (64 KiB, two workgroups per CU): a tile's loads have ONE iteration to land. NW = 8, NBUF = 4 (128 KiB, one workgroup of
  // 256 queries per CU): three iterations, and half the DMA traffic per CU (ablation: the 2-stage form lost 15 % to the loads' wait).
  __shared__ __attribute__((aligned(1024))) unsigned char ring[2 * NBUF * STAGE];      // K stages | V stages
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
  const int ntiles = (a.T + 63) >> 6;

  // LDS image of a stage (cdna_hip_programming.md T10, image (a)): 16-B chunk ch of row `row` at
  //   2048 (row >> 3) + 512 (ch >> 2) + 64 (row & 7) + 16 ((ch & 3) ^ ((row >> 2) & 3))
  // - every fragment read below is (one of four per-lane bases) + a compile-time constant, so the reads cost no address arithmetic.
  // DMA: instruction n = (row block n >> 1, chunk-group pair n & 1) fills two 512-B subtiles (1 KiB): lane -> subtile lane >> 5, row
  // (lane & 31) >> 2 of the block, physical chunk lane & 3. A whole tile is (wave-uniform tile pointer) + (per-lane byte offset fixed for
  // the kernel): one address add per instruction.
  constexpr int NI = 16 / NW;                    // instructions per wave, tile and operand
  const int dsub = lane >> 5, drr = (lane & 31) >> 2, dpc = lane & 3;
  unsigned doff[NI];
  bool dok[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int n = wave * NI + i, row = 8 * (n >> 1) + drr;
    const int ch = 4 * (2 * (n & 1) + dsub) + (dpc ^ ((row >> 2) & 3));
    doff[i] = (unsigned)((row * C3 + ch * 8) * 2);
    dok[i] = ch * 8 < CH;
  }
  auto issue = [&](int kt, const int which /* 1: K rows, 2: V rows */) __attribute__((always_inline)) {
    unsigned char* dst = ring + (which == 2 ? NBUF * STAGE : 0) + (kt % NBUF) * STAGE + wave * (NI * 1024);
    if (kt * 64 + 64 <= a.T) {
      const unsigned char* tp = reinterpret_cast<const unsigned char*>(base + (long)kt * 64 * C3 + which * CH);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (dok[i]) GLDS16(tp + doff[i], dst + i * 1024);
        else if (which == 1 && CH < DP) GLDS16(g_attn_zero + (lane & 15) * 16, dst + i * 1024);   // zero extension of K, ch = 16 (V's extension rows feed output channels that are never stored)
      }
    } else {
#pragma unroll
      for (int i = 0; i < NI; ++i) {              // last, partial tile: rows beyond T re-read row T - 1 (their logits are masked)
        const int n = wave * NI + i, row = 8 * (n >> 1) + drr;
        const int ch = 4 * (2 * (n & 1) + dsub) + (dpc ^ ((row >> 2) & 3));
        const T* src = base + (long)min(kt * 64 + row, a.T - 1) * C3 + ch * 8 + which * CH;
        if (dok[i]) GLDS16(src, dst + i * 1024);
        else if (which == 1 && CH < DP) GLDS16(g_attn_zero + (lane & 15) * 16, dst + i * 1024);
      }
    }
  };
  // One block barrier per iteration kt = -1 .. ntiles - 1. Loads leave in the order they are needed: K(0), then the pairs m = 0, 1, ..:
  // {K(m + 1), V(m)} (what iteration m multiplies), NBUF - 1 pairs ahead. After the barrier of iteration kt pair kt is complete in LDS
  // (counted vmcnt: the NBUF - 2 younger pairs may still be in flight) and every wave is done with K(kt) and V(kt - 1), whose stages take
  // pair kt + NBUF - 1 = {K(kt + NBUF), V(kt + NBUF - 1)}.
  auto top = [&](int kt) __attribute__((always_inline)) {
    // counted only where every DMA instruction has live lanes (CH = 128: narrower heads leave whole instructions without a lane)
    if (CH == 128 && (kt < 0 ? NBUF - 1 : kt + NBUF - 1) < ntiles) {      // every younger pair was issued whole (2 NI instructions each; pair -1 is K(0) alone, all NBUF - 1 pairs younger)
      if (kt < 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI * (NBUF - 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI * (NBUF - 2)) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (!AT_DIAG(8)) __builtin_amdgcn_s_barrier();
    if constexpr (AT_DIAG(4)) return;
    if (kt < 0) return;
    if (kt + NBUF < ntiles) issue(kt + NBUF, 1);
    if (kt + NBUF - 1 < ntiles) issue(kt + NBUF - 1, 2);
  };
  issue(0, 1);                                   // the first loads leave together: one memory latency in front of the first product
#pragma unroll
  for (int m = 0; m < NBUF - 1; ++m) {
    if (m + 1 < ntiles) issue(m + 1, 1);
    if (m < ntiles) issue(m, 2);
  }
  if (!live) {                                   // a wave without queries only feeds the rings
    for (int kt = -1; kt < ntiles; ++kt) top(kt);
    return;
  }
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
  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
  // per-lane bases of the fragment reads (bytes inside a stage)
  const unsigned rq = (unsigned)((r >> 2) & 3), lds0 = (unsigned)(uintptr_t)ring;
  const unsigned kb0 = lds0 + 2048u * (r >> 3) + 64u * (r & 7) + 16u * ((unsigned)h ^ rq);            // K row read, k-step s even: + 8192 it + 512 (s >> 1)
  const unsigned kb1 = lds0 + 2048u * (r >> 3) + 64u * (r & 7) + 16u * ((2u + (unsigned)h) ^ rq);     //             k-step s odd
  const unsigned g1 = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;                    // transposed read: lane 4 q + p of its 16-lane group
  const unsigned vb0 = lds0 + 64u * (4 * h + tq) + 16u * ((2 * g1 + (tp >> 1)) ^ (unsigned)h) + 8u * (tp & 1);          // rows 4 h + q      of the 16-key block: + 2048 (4 it + 2 u) + 512 dt
  const unsigned vb1 = lds0 + 64u * (4 * h + tq) + 16u * ((2 * g1 + (tp >> 1)) ^ (unsigned)(h + 2)) + 8u * (tp & 1);    // rows 4 h + q + 8: + 2048 more
  // Softmax without a per-tile row maximum (the scheme of lsa_flash64_kernel, svit.hip): the logits leave the MFMA already relative to a
  // per-query reference m_ref - one more k-step whose K side is the constant 1 and whose Q side is -m_ref - so p = exp2(c s'). The reference is
  // the first tile's row maximum and moves only when a tile's probabilities outgrow what the operand type and the fp32 sums carry (their own
  // row sum, which the loop has anyway, is the test): then O, l and the logits at hand go to the new reference. Any reference gives the same
  // softmax; this one is never above the running row maximum, so nothing underflows that a per-tile maximum would have kept.
  V8 ka, qa;
#pragma unroll
  for (int j = 0; j < 8; ++j) { ka[j] = (T)0.f; qa[j] = (T)0.f; }
  if (h == 0) ka[0] = (T)1.f;
  float m_ref = 0.f, l_run = 0.f;                // m_ref in logit units; l_run: this lane half's keys only (the halves meet in the epilogue)
  constexpr float kLimit = __is_same(T, _Float16) ? 16384.f : 1.2089258e24f;
  const f32x2 c2 = {a.c, a.c};
  auto mask_tail = [&](f32x16 (&s)[2], int kt) __attribute__((always_inline)) {
    if (kt * 64 + 64 > a.T) {
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (kt * 64 + 32 * it + (e & 3) + 8 * (e >> 2) + 4 * h >= a.T) s[it][e] = -INFINITY;
    }
  };
  // p = exp2(c s') of elements [E0, E1) (pairs): row sum, every completed octet packed to the operand type (s' stays intact for a move of the reference)
#define AT_SLICE(S, E0, E1)                                                                      \
  _Pragma("unroll") for (int e_ = (E0); e_ < (E1); e_ += 2) {                                   \
    const f32x2 x2_ = f32x2{S[e_ >> 4][e_ & 15], S[e_ >> 4][(e_ & 15) + 1]} * c2;              \
    pt[e_ & 7] = __builtin_amdgcn_exp2f(x2_[0]);                                                 \
    pt[(e_ & 7) + 1] = __builtin_amdgcn_exp2f(x2_[1]);                                           \
    sum += pt[e_ & 7];                                                                           \
    sum += pt[(e_ & 7) + 1];                                                                     \
    if ((e_ & 7) == 6) {                                                                         \
      _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) pb[e_ >> 3][j_] = (T)pt[j_];             \
      asm volatile("" : "+v"(pb[e_ >> 3]));   /* (pin: see below) */                             \
    }                                                                                            \
  }                                                                                              \
  /* pin the slice where it stands: its results are only read after the products, and the optimizer would sink all of it behind them */ \
  asm volatile("" : "+v"(sum));
  // move of the reference to (at least) the row maximum of the logits in `sc`: O, l, sc and the next tile's logits `sn` follow
  auto move_ref = [&](f32x16 (&sc)[2], f32x16* sn, const bool first) __attribute__((always_inline)) {
    float mx = sc[0][0];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sc[it][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float up = first ? fmaxf(mx, -30000.f) : fmaxf(mx, 0.f);
    const float m_new = (float)(T)(m_ref + up);
    const float delta = m_new - m_ref;
    const float alpha = __builtin_amdgcn_exp2f(-delta * a.c);
    m_ref = m_new;
    if (h == 0) qa[0] = (T)(-m_new);
    l_run *= alpha;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int e = 0; e < 16; ++e) { sc[it][e] -= delta; if (sn) sn[it][e] -= delta; }
  };
  // ---- iteration kt >= 0 (PAR = kt & 1 selects the stages at compile time): S'(kt + 1) = K(kt + 1) Q^T - m_ref on the matrix pipe while the
  //      vector pipe exponentiates S'(kt); then O^T += V(kt)^T P(kt). One MFMA, then its share of the vector work, fenced by sched_barrier:
  //      an in-order wave overlaps the two pipes only if its instruction stream alternates (the other wave of the SIMD fills the second
  //      product's vector-idle gaps with its first). The last iteration computes a product nobody reads (from the stale stage of K(kt - 1)):
  //      18 of a workgroup's 34 x ntiles MFMAs, against a second copy of the loop body and its register pressure.
  auto iter = [&](auto par_c, f32x16 (&sc)[2], f32x16 (&sn)[2], const int kt) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;      // kt % NBUF
    constexpr unsigned KST = (unsigned)(((PAR + 1) % NBUF) * STAGE), VST = (unsigned)(NBUF * STAGE + PAR * STAGE);
    // (the DS offset field has 16 bits: the stage goes into the four bases, one add each per iteration)
    const unsigned kb0s = kb0 + KST, kb1s = kb1 + KST, vb0s = vb0 + VST, vb1s = vb1 + VST;
    // Operand fragments: groups of G = 2 in a ring of three register sets, read two groups ahead of the MFMAs that consume them, by inline
    // asm with hand-counted s_waitcnt lgkmcnt (the compiler's own counting put lgkmcnt(0) behind every prefetch: profiles/r05_attn_flash.md).
    // K group g (MFMAs 2 g, 2 g + 1 of the first product) sits in set g % 3, V group v (second product) in set (NGK + v) % 3: the first two V
    // groups leave with the last K group's wait, so the second product starts without an LDS latency of its own.
    constexpr int G = 2, NGK = NQK / G, NGV = NPV / G;
    V8 fr[3][G];
    s16x4 vl[3][G], vh[3][G];
    auto kissue = [&](auto g_c) __attribute__((always_inline)) {
      constexpr int g = decltype(g_c)::value;
      (void)&fr; (void)kb0s; (void)kb1s;          // (operands of inline asm alone do not capture in a generic lambda)
#pragma unroll
      for (int jj = 0; jj < G; ++jj) {
        // MFMA j = 2 s + it: row block it, k-step s -> base kb(s & 1), constant 8192 it + 512 (s >> 1)
        if constexpr (AT_DIAG(2)) { asm volatile("" : "=v"(fr[g % 3][jj])); continue; }
        if (jj == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[g % 3][0]) : "v"(((g * G) >> 1) & 1 ? kb1s : kb0s), "n"(((g * G) & 1) * 8192 + ((g * G) >> 2) * 512));
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[g % 3][1]) : "v"(((g * G + 1) >> 1) & 1 ? kb1s : kb0s), "n"(((g * G + 1) & 1) * 8192 + ((g * G + 1) >> 2) * 512));
      }
    };
    auto vissue = [&](auto v_c) __attribute__((always_inline)) {
      constexpr int v = decltype(v_c)::value, st = (NGK + v) % 3;
      (void)&vl; (void)&vh; (void)vb0s; (void)vb1s;
      if constexpr (AT_DIAG(2)) { asm volatile("" : "=v"(vl[st][0]), "=v"(vh[st][0]), "=v"(vl[st][1]), "=v"(vh[st][1])); return; }
      // MFMA j: (it, u) = j / DT, dt = j % DT -> constant 4096 (j / DT) + 512 (j % DT); rows + 8 of the 16-key block: + 2048
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vl[st][0]) : "v"(vb0s), "n"(4096 * ((v * G) / DT) + 512 * ((v * G) % DT)));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vh[st][0]) : "v"(vb1s), "n"(4096 * ((v * G) / DT) + 512 * ((v * G) % DT) + 2048));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vl[st][1]) : "v"(vb0s), "n"(4096 * ((v * G + 1) / DT) + 512 * ((v * G + 1) % DT)));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vh[st][1]) : "v"(vb1s), "n"(4096 * ((v * G + 1) / DT) + 512 * ((v * G + 1) % DT) + 2048));
    };
    kissue(std::integral_constant<int, 0>{});
    if constexpr (NGK > 1) kissue(std::integral_constant<int, 1>{});
    V8 pb[4];
    float pt[8];
    float sum = 0.f;
    constexpr int NSLOT = NQK + 2;
    static_for<0, NSLOT>([&](auto sl_c) __attribute__((always_inline)) {
      constexpr int sl = decltype(sl_c)::value;
      (void)&fr; (void)&pb; (void)&pt; (void)&sum; (void)&sc; (void)&sn;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (sl < 2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) sn[sl][e] = 0.f;
        if constexpr (!AT_DIAG(16)) sn[sl] = MM<T>::mfma(ka, qa, sn[sl]);
      } else {
        constexpr int j = sl - 2, g = j / G;
        if constexpr (j % G == 0) {
          // two groups ahead; the last K group sends the first two V groups instead
          if constexpr (g + 2 < NGK) kissue(std::integral_constant<int, g + 2>{});
          if constexpr (g == NGK - 1) { vissue(std::integral_constant<int, 0>{}); if constexpr (NGV > 1) vissue(std::integral_constant<int, 1>{}); }
          constexpr int younger = g == NGK - 1 ? (NGV > 1 ? 8 : 4) : (g + 2 < NGK ? 2 * G : (g + 1 < NGK ? G : 0)) + (g == NGK - 2 ? 0 : 0);
          asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fr[g % 3][0]), "+v"(fr[g % 3][1]) : "n"(younger));
        }
        if constexpr (!AT_DIAG(16)) sn[j & 1] = MM<T>::mfma(fr[g % 3][j % G], qf[j >> 1], sn[j & 1]);
        else asm volatile("" : "+v"(sn[j & 1]) : "v"(fr[g % 3][j % G]));
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!AT_DIAG(1)) { AT_SLICE(sc, ((sl * 16) / NSLOT) * 2, (((sl + 1) * 16) / NSLOT) * 2) }
    });
    if constexpr (AT_DIAG(1)) { _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) asm volatile("" : "=v"(pb[k_])); sum = 0.f; asm volatile("" : "+v"(sum)); }
    __builtin_amdgcn_sched_barrier(0);
    mask_tail(sn, kt + 1);
    if (__builtin_amdgcn_ballot_w64(!(sum <= kLimit))) {       // rare: the tile outgrew the reference (an overflowed exponential makes the sum inf, a NaN fails the comparison too)
      move_ref(sc, sn, false);
      sum = 0.f;
      AT_SLICE(sc, 0, 32)
    }
    l_run += sum;
    static_for<0, NPV>([&](auto j_c) __attribute__((always_inline)) {
      constexpr int j = decltype(j_c)::value, v = j / G, st = (NGK + v) % 3;
      (void)&vl; (void)&vh; (void)&pb;
      if constexpr (j % G == 0) {
        if constexpr (v + 2 < NGV) vissue(std::integral_constant<int, v + 2>{});
        constexpr int younger = (v + 2 < NGV ? 8 : (v + 1 < NGV ? 4 : 0));
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(vl[st][0]), "+v"(vh[st][0]), "+v"(vl[st][1]), "+v"(vh[st][1]) : "n"(younger));
      }
      const s16x8 v8 = __builtin_shufflevector(vl[st][j % G], vh[st][j % G], 0, 1, 2, 3, 4, 5, 6, 7);
      if constexpr (!AT_DIAG(16)) o[j % DT] = MM<T>::mfma(__builtin_bit_cast(V8, v8), pb[j / DT], o[j % DT]);
      else asm volatile("" : "+v"(o[j % DT]) : "v"(v8), "v"(pb[j / DT]));
    });
  };

  f32x16 sa[2], sb[2];
  // ---- iteration -1: S'(0), and the first reference = its row maximum
  top(-1);
  {
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int e = 0; e < 16; ++e) sa[it][e] = 0.f;
#pragma unroll
    for (int j = 0; j < NQK; ++j)
      sa[j & 1] = MM<T>::mfma(*reinterpret_cast<const V8*>(ring + ((((j >> 1) & 1) ? kb1 : kb0) - lds0) + (j & 1) * 8192 + (j >> 2) * 512), qf[j >> 1], sa[j & 1]);
    mask_tail(sa, 0);
    move_ref(sa, nullptr, true);
  }
#pragma unroll 1
  for (int kt = 0; kt < ntiles; kt += NBUF) {
    top(kt);
    iter(std::integral_constant<int, 0>{}, sa, sb, kt);
    if (kt + 1 < ntiles) {
      top(kt + 1);
      iter(std::integral_constant<int, 1>{}, sb, sa, kt + 1);
    }
    if constexpr (NBUF == 4) {
      if (kt + 2 < ntiles) {
        top(kt + 2);
        iter(std::integral_constant<int, 2>{}, sa, sb, kt + 2);
      }
      if (kt + 3 < ntiles) {
        top(kt + 3);
        iter(std::integral_constant<int, 3>{}, sb, sa, kt + 3);
      }
    }
  }
#undef AT_SLICE
  // lane (query r) holds channels d = 32 dt + (e & 3) + 8 (e >> 2) + 4 h: 4 consecutive channels per e-quad -> 8-B stores
  l_run += __shfl_xor(l_run, 32, 64);
  if (q0 + r >= a.T) return;
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

template <typename T, int NW, int NBUF>
static int attn_flash_launch(AttnFlashArgs a, int ch, hipStream_t st) {
  static_assert(NBUF == 2 || NBUF == 4, "ring depth");
  a.nq = (a.T + NW * 32 - 1) / (NW * 32);
  const int grid = a.nbh * a.nq;
  if (ch == 128) attn_flash_kernel<T, 128, NW, NBUF><<<grid, NW * 64, 0, st>>>(a);
  else if (ch == 64) attn_flash_kernel<T, 64, NW, NBUF><<<grid, NW * 64, 0, st>>>(a);
  else if (ch == 32) attn_flash_kernel<T, 32, NW, NBUF><<<grid, NW * 64, 0, st>>>(a);
  else attn_flash_kernel<T, 16, NW, NBUF><<<grid, NW * 64, 0, st>>>(a);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_attn_legacy16(const void* qkv, int qkv_is16, void* out16, int B, int T, int heads, int ch, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(qkv && out16 && B > 0 && heads > 0, "attn_legacy16: bad args");
  STEDM_CHECK_ARG(T > 0 && (ch == 128 || ch == 64 || ch == 32 || ch == 16) && ((T == 64 && ch != 16) || qkv_is16),
                  "attn_legacy16: covers head widths 16 / 32 / 64 / 128 from the 16-bit qkv plane (any T) and T == 64 (ch >= 32) from fp32 rows (T=%d ch=%d is16=%d)", T, ch, qkv_is16);
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "attn_legacy16: bad mm_dtype");
  if (T != 64 || ch == 16) {
    hipStream_t s_ = as_stream(stream);
    AttnFlashArgs fa{qkv, out16, T, heads, B * heads, (T + 127) / 128, 1.4426950408889634f / sqrtf((float)ch)};
    STEDM_CHECK_ARG((long)64 * heads * 3 * ch * 2 < (1l << 31), "attn_legacy16: row too wide");
    // 256-query workgroups (one per CU, four-stage rings) for long sequences of 128-wide heads when they fill the chip (+2.5 % at T = 1024,
    // +3.5 % at 4096; slower at T = 256 and at ch = 64: profiles/r05_attn_flash.md); otherwise 128-query workgroups, two per CU
    static int cus = 0;
    if (cus == 0) { cus = stedm_device_cus(); if (cus <= 0) cus = 256; }
    static const int force = getenv("STEDM_ATTN_FORM") ? atoi(getenv("STEDM_ATTN_FORM")) : 0;      // A/B timing only: 4 / 8 waves
    const bool wide = force ? force == 8 : (T >= 1024 && ch == 128 && (long)B * heads * ((T + 255) / 256) >= cus);
    if (wide) return mm_dtype == STEDM_F16 ? attn_flash_launch<_Float16, 8, 4>(fa, ch, s_) : attn_flash_launch<__bf16, 8, 4>(fa, ch, s_);
    return mm_dtype == STEDM_F16 ? attn_flash_launch<_Float16, 4, 2>(fa, ch, s_) : attn_flash_launch<__bf16, 4, 2>(fa, ch, s_);
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
