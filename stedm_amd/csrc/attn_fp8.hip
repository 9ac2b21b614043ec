// fp8 (OCP e4m3) variant of the style encoder's LSA flash attention (BASELINE config 5: "fp8 MFMA attention"; vit_set.py:52-67).
// Same algorithm and tiling as lsa_flash_kernel (svit.hip): one block = 128 queries of one (sample, head), 4 waves x 32 queries,
// keys / values in tiles of 64, S^T = K Q^T with the softmax in the accumulators, O^T += V^T P^T with P taken from the accumulators
// as the next MFMA's operand — but every MFMA operand is e4m3 (v_mfma_f32_32x32x16_fp8_fp8, fp32 accumulate):
//   q, k, v : per-TENSOR scales 448 / amax (amax over the whole [B][T][H*64] block of q / k / v, found by stedm_qkv_amax);
//   P       : exp2(s - max) in [0, 1] times 256 (e4m3 holds up to 448);
//   logits, running max / sum, the rescale and the normalisation are fp32, as in every other mode.
// Half the operand bytes of the 16-bit kernel through HBM, LDS and the registers; the non-scaled fp8 MFMA runs at the bf16 rate
// (MI355X_MICROARCH.md, Matrix cores), so this is a bandwidth / capacity mode, and a precision experiment: its deviation from the
// reference is measured and reported (tests/test_gpu_style.py), never asserted at 1e-3.
#include <float.h>

#include "conv_common.hpp"
using namespace stedm;

namespace {

__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);   // bytes 0, 1
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);    // bytes 2, 3
  return (unsigned)w;
}

// amax[0..2] = max |q * qscale|, max |k|, max |v| over qkv fp32 [M][3*HD] (HD = heads * 64): float bits of non-negative values order
// like unsigned integers, so the block maxima are combined with atomicMax on the bit pattern (the result does not depend on order)
__global__ void __launch_bounds__(256) qkv_amax_kernel(const float* __restrict__ qkv, float qscale, long M, int HD, unsigned* __restrict__ amax) {
  __shared__ float red[3][4];
  float m[3] = {0.f, 0.f, 0.f};
  const long total = M * 3 * HD;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int which = (int)((i % (3 * HD)) / HD);
    const float v = fabsf(qkv[i]) * (which == 0 ? fabsf(qscale) : 1.0f);
    m[0] = which == 0 ? fmaxf(m[0], v) : m[0];
    m[1] = which == 1 ? fmaxf(m[1], v) : m[1];
    m[2] = which == 2 ? fmaxf(m[2], v) : m[2];
  }
#pragma unroll
  for (int w = 0; w < 3; ++w) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m[w] = fmaxf(m[w], __shfl_xor(m[w], o, 64));
    if ((threadIdx.x & 63) == 0) red[w][threadIdx.x >> 6] = m[w];
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const float v = fmaxf(fmaxf(red[threadIdx.x][0], red[threadIdx.x][1]), fmaxf(red[threadIdx.x][2], red[threadIdx.x][3]));
    atomicMax(amax + threadIdx.x, __float_as_uint(v));
  }
}

// qkv fp32 [B][T][3*H*64] -> q8 / k8 [B*H][Tp][64] bytes (rows >= T zero), vT8 [B*H][64][Tp] (cols >= T zero), each scaled by 448 / amax
__global__ void __launch_bounds__(256) qkv_pack_fp8_kernel(const float* __restrict__ qkv, float qscale, const float* __restrict__ amax,
                                                           unsigned char* __restrict__ q8, unsigned char* __restrict__ k8, unsigned char* __restrict__ v8,
                                                           int Tn, int Tp, int H) {
  __shared__ float sv[64][65];
  const int bh = blockIdx.x, b = bh / H, hd = bh % H;
  const int t0 = blockIdx.y * 64;
  const int HD = H * 64;
  const float sq = 448.0f / fmaxf(amax[0], 1e-20f) * qscale, sk = 448.0f / fmaxf(amax[1], 1e-20f), sv_ = 448.0f / fmaxf(amax[2], 1e-20f);
  for (int i = threadIdx.x; i < 64 * 16; i += 256) {          // one thread = 4 consecutive channels of a token
    const int tl = i >> 4, d = (i & 15) * 4;
    const int t = t0 + tl;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f), k = q, v = q;
    if (t < Tn) {
      const float* p = qkv + ((long)b * Tn + t) * (3 * HD) + hd * 64 + d;
      q = *reinterpret_cast<const float4*>(p); k = *reinterpret_cast<const float4*>(p + HD); v = *reinterpret_cast<const float4*>(p + 2 * HD);
    }
    const long o = ((long)bh * Tp + t) * 64 + d;
    *reinterpret_cast<unsigned*>(q8 + o) = pack4_fp8(q.x * sq, q.y * sq, q.z * sq, q.w * sq);
    *reinterpret_cast<unsigned*>(k8 + o) = pack4_fp8(k.x * sk, k.y * sk, k.z * sk, k.w * sk);
    sv[tl][d] = v.x * sv_; sv[tl][d + 1] = v.y * sv_; sv[tl][d + 2] = v.z * sv_; sv[tl][d + 3] = v.w * sv_;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 16; i += 256) {          // one thread = 4 consecutive positions of a channel
    const int d = i >> 4, tl = (i & 15) * 4;
    // V^T keys permuted inside every group of 16 (order 0-3, 8-11, 4-7, 12-15: the k order of the accumulator-as-operand tile), so that a
    // lane's 8 values are one contiguous 8-B read: the 4 positions of this thread hold 4 consecutive keys starting at kq
    const int p16 = tl & 15, kq = (tl & ~15) | ((p16 & 3) | ((p16 & 4) << 1) | ((p16 & 8) >> 1));
    const long o = ((long)bh * 64 + d) * Tp + t0 + tl;
    *reinterpret_cast<unsigned*>(v8 + o) = pack4_fp8(sv[kq][d], sv[kq + 1][d], sv[kq + 2][d], sv[kq + 3][d]);
  }
}

struct Flash8Args {
  const unsigned char *q, *k, *v;
  const float* amax;
  void* out;     // [B][T][H*64] 16-bit plane
  int T, Tp, H;
};

template <typename T>
__global__ void __launch_bounds__(256, 2) lsa_flash_fp8_kernel(Flash8Args a) {
  constexpr int RS = 80;                           // LDS row stride in bytes (64 + 16 pad)
  constexpr int TILE = 64 * RS;
  constexpr int O_BYTES = 4 * 32 * 65 * (int)sizeof(float);
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[2 * TILE > O_BYTES ? 2 * TILE : O_BYTES];
  unsigned char* sK = lds_raw;
  unsigned char* sV = lds_raw + TILE;
  float (*sO)[32][65] = reinterpret_cast<float (*)[32][65]>(lds_raw);
  // XCD-aware block -> (sample-head, query tile): the blocks of one sample-head all run on one XCD (block id mod 8), so its K / V^T stay in
  // that XCD's L2 (see lsa_flash_dma_kernel, svit.hip)
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
  const float s_logit = (fmaxf(a.amax[0], 1e-20f) / 448.0f) * (fmaxf(a.amax[1], 1e-20f) / 448.0f);   // fp8 q.k -> log2-domain logit
  const float s_out = (fmaxf(a.amax[2], 1e-20f) / 448.0f) / 256.0f;                                   // (256 P) (448 / amax_v v) -> P v

  long qf[4];     // the wave's 32 queries: lane (r, h) holds q[q0 + r][16 ks + 8 h .. + 7] for k-step ks
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const long*>(a.q + ((long)bh * a.Tp + q0 + r) * 64 + ks * 16 + h * 8);

  f32x16 o[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int qidx = q0 + r;
  const int ntiles = (a.T + 63) / 64;
  // 64 x 64 B per tile and operand: 256 threads x 16 B; next tile prefetched through registers
  const int frow = tid >> 2, fc = tid & 3;
  uint4 kr, vr;
#define FETCH8(KT)                                                                                             \
  {                                                                                                            \
    kr = *reinterpret_cast<const uint4*>(a.k + ((long)bh * a.Tp + (KT) * 64 + frow) * 64 + fc * 16);           \
    vr = *reinterpret_cast<const uint4*>(a.v + ((long)bh * 64 + frow) * a.Tp + (KT) * 64 + fc * 16);           \
  }
  FETCH8(0)
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();
    *reinterpret_cast<uint4*>(sK + frow * RS + fc * 16) = kr;
    *reinterpret_cast<uint4*>(sV + frow * RS + fc * 16) = vr;
    __syncthreads();
    if (kt + 1 < ntiles) FETCH8(kt + 1)
    f32x16 s[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[sub][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const long kf = *reinterpret_cast<const long*>(sK + (sub * 32 + r) * RS + ks * 16 + h * 8);
        s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(kf, qf[ks], s[sub], 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) s[sub][e] *= s_logit;
    }
    if ((q0 >> 6) == kt) {      // the diagonal: a token never attends to itself (vit_set.py:58-60)
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
    // ---- O^T += V^T P^T: P registers 8 s2 .. 8 s2 + 7 of a 32-key sub-tile are the 8 k-slots of k-step s2 (same k permutation as the
    // 16-bit kernel: slot j of lane half h is key 16 s2 + 8 (j >> 2) + 4 h + (j & 3)), V^T is gathered in that order
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const unsigned plo = pack4_fp8(s[sub][8 * s2] * 256.f, s[sub][8 * s2 + 1] * 256.f, s[sub][8 * s2 + 2] * 256.f, s[sub][8 * s2 + 3] * 256.f);
        const unsigned phi = pack4_fp8(s[sub][8 * s2 + 4] * 256.f, s[sub][8 * s2 + 5] * 256.f, s[sub][8 * s2 + 6] * 256.f, s[sub][8 * s2 + 7] * 256.f);
        const long pf = (long)(((unsigned long)phi << 32) | plo);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const long vf = *reinterpret_cast<const long*>(sV + (d * 32 + r) * RS + sub * 32 + s2 * 16 + h * 8);
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(vf, pf, o[d], 0, 0, 0);
        }
      }
  }
#undef FETCH8
  __syncthreads();
  const float inv = s_out / l_run;
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) sO[wave][r][d * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] = o[d][e] * inv;
  __syncthreads();
  {
    const int row = lane >> 1, half = lane & 1;
    const int t = q0 + row;
    if (t < a.T) {
      T* oh = reinterpret_cast<T*>(a.out) + ((long)b * a.T + t) * (a.H * 64) + hd * 64 + half * 32;
#pragma unroll
      for (int j = 0; j < 32; ++j) oh[j] = (T)sO[wave][row][half * 32 + j];
    }
  }
}

}  // namespace

extern "C" int stedm_qkv_amax(const float* qkv, float qscale, long M, int heads, float* amax, void* stream) {
  STEDM_CHECK_ARG(qkv && amax && M > 0 && heads > 0, "qkv_amax: bad args");
  hipStream_t st = as_stream(stream);
  STEDM_HIP_TRY(hipMemsetAsync(amax, 0, 3 * sizeof(float), st));
  const long total = M * 3 * heads * 64;
  const int grid = (int)((total + 256 * 16 - 1) / (256 * 16) < 2048 ? (total + 256 * 16 - 1) / (256 * 16) : 2048);
  qkv_amax_kernel<<<grid, 256, 0, st>>>(qkv, qscale, M, heads * 64, reinterpret_cast<unsigned*>(amax));
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_qkv_pack_fp8(const float* qkv, float qscale, const float* amax, void* q8, void* k8, void* vt8, int B, int T, int Tp, int heads,
                                  void* stream) {
  STEDM_CHECK_ARG(qkv && amax && q8 && k8 && vt8, "qkv_pack_fp8: null pointer");
  STEDM_CHECK_ARG(Tp % 128 == 0 && Tp >= T, "qkv_pack_fp8: Tp must be a multiple of 128 and >= T");
  dim3 grid(B * heads, Tp / 64);
  qkv_pack_fp8_kernel<<<grid, 256, 0, as_stream(stream)>>>(qkv, qscale, amax, (unsigned char*)q8, (unsigned char*)k8, (unsigned char*)vt8, T, Tp, heads);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_lsa_flash_fp8(const void* q8, const void* k8, const void* vt8, const float* amax, void* out16, int B, int T, int Tp, int heads,
                                   int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(q8 && k8 && vt8 && amax && out16, "lsa_flash_fp8: null pointer");
  STEDM_CHECK_ARG(Tp % 128 == 0 && Tp >= T && T > 1, "lsa_flash_fp8: need Tp %% 128 == 0, Tp >= T > 1");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "lsa_flash_fp8: bad mm_dtype %d (type of the output plane)", mm_dtype);
  Flash8Args a{(const unsigned char*)q8, (const unsigned char*)k8, (const unsigned char*)vt8, amax, out16, T, Tp, heads};
  dim3 grid(B * heads, Tp / 128);
  if (mm_dtype == STEDM_F16) lsa_flash_fp8_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>(a);
  else lsa_flash_fp8_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>(a);
  STEDM_LAUNCH_CHECK();
  return 0;
}
