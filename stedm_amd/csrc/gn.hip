// GroupNorm statistics -> per-(sample, channel) scale/shift, over the virtual concat [x1 | x2].
// One block per (sample, group): two passes over its cpg x HW slice (second pass hits L1/L2),
// so the variance is the centred, biased form nn.GroupNorm computes (util.py:214-216).
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
using namespace stedm;

struct GnArgs {
  const float* x1;
  const float* x2;
  int c1, c2, bmod, groups, HW;
  const float* gamma;
  const float* beta;
  float eps;
  float* scale;
  float* shift;
};

template <bool VEC4>
__global__ void __launch_bounds__(256) gn_scale_shift_kernel(GnArgs a) {
  __shared__ float red[4];
  const int b = blockIdx.x / a.groups, g = blockIdx.x % a.groups;
  const int C = a.c1 + a.c2;
  const int cpg = C / a.groups;
  const int cbeg = g * cpg;
  const float* p1 = a.x1 + (long)b * a.HW * a.c1;
  const float* p2 = a.x2 ? a.x2 + (long)(a.bmod > 0 ? b % a.bmod : b) * a.HW * a.c2 : nullptr;
  const int n = cpg * a.HW;

  auto load1 = [&](int e) -> float {
    const int pix = e / cpg, c = cbeg + (e - pix * cpg);
    return c < a.c1 ? p1[(long)pix * a.c1 + c] : p2[(long)pix * a.c2 + (c - a.c1)];
  };
  auto load4 = [&](int e4) -> float4 {  // e4 indexes quads; cpg % 4 == 0 and c1 % 4 == 0
    const int qpg = cpg >> 2;
    const int pix = e4 / qpg, c = cbeg + ((e4 - pix * qpg) << 2);
    return c < a.c1 ? *reinterpret_cast<const float4*>(p1 + (long)pix * a.c1 + c)
                    : *reinterpret_cast<const float4*>(p2 + (long)pix * a.c2 + (c - a.c1));
  };

  float s = 0.f;
  if (VEC4) {
    for (int e = threadIdx.x; e < (n >> 2); e += 256) {
      const float4 v = load4(e);
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (int e = threadIdx.x; e < n; e += 256) s += load1(e);
  }
  const float mean = block_sum_256(s, red) / (float)n;
  float q = 0.f;
  if (VEC4) {
    for (int e = threadIdx.x; e < (n >> 2); e += 256) {
      const float4 v = load4(e);
      const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
      q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  } else {
    for (int e = threadIdx.x; e < n; e += 256) {
      const float d = load1(e) - mean;
      q += d * d;
    }
  }
  const float var = block_sum_256(q, red) / (float)n;
  const float rstd = 1.0f / sqrtf(var + a.eps);
  for (int c = threadIdx.x; c < cpg; c += 256) {
    const float gm = a.gamma[cbeg + c] * rstd;
    a.scale[(long)b * C + cbeg + c] = gm;
    a.shift[(long)b * C + cbeg + c] = a.beta[cbeg + c] - mean * gm;
  }
}

extern "C" int stedm_gn_scale_shift(const float* x1, int c1, const float* x2, int c2, int x2_bmod, const float* gamma,
                                    const float* beta, float eps, int groups, int B, int HW, float* scale, float* shift,
                                    void* stream) {
  STEDM_CHECK_ARG(x1 && gamma && beta && scale && shift, "gn_scale_shift: null pointer");
  STEDM_CHECK_ARG((x2 != nullptr) == (c2 > 0), "gn_scale_shift: x2/c2 mismatch");
  const int C = c1 + c2;
  STEDM_CHECK_ARG(groups > 0 && C % groups == 0, "gn_scale_shift: C=%d not divisible by groups=%d", C, groups);
  STEDM_CHECK_ARG(B > 0 && HW > 0, "gn_scale_shift: bad B/HW");
  GnArgs a{x1, x2, c1, c2, x2_bmod, groups, HW, gamma, beta, eps, scale, shift};
  const int cpg = C / groups;
  const bool vec = (cpg % 4 == 0) && (c1 % 4 == 0) && (c2 % 4 == 0);
  if (vec)
    gn_scale_shift_kernel<true><<<B * groups, 256, 0, as_stream(stream)>>>(a);
  else
    gn_scale_shift_kernel<false><<<B * groups, 256, 0, as_stream(stream)>>>(a);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// ================================================================================================
// Two-kernel GroupNorm for the DMA convolution path
// ================================================================================================
// (1) stats: block = (sample, slab of pixels); threads sweep whole NHWC rows with 16-B loads (coalesced),
//     accumulate per-group partial sums in LDS and add them to stats[b][g] = {sum, sumsq} with fp64 atomics.
struct GnStatsArgs {
  const float* x1;
  const float* x2;
  int c1, c2, bmod, groups, HW, slab;
  double* stats;
};

constexpr int GN_MAX_SLOTS = 4;   // channel quads owned by one thread (C <= 4096)

__global__ void __launch_bounds__(256) gn_stats_kernel(GnStatsArgs a) {
  // Every thread owns fixed channel quads (so its running sums never change group) and strides over the slab's
  // pixels; per-channel partials go to LDS and one thread per group adds them in a fixed order: no atomics, the
  // result is bitwise reproducible.
  extern __shared__ float part[];   // [slots][256][8] : {sum x4, sumsq x4} per thread and slot
  const int b = blockIdx.x, slab = blockIdx.y;
  const int C = a.c1 + a.c2, Q = C >> 2;
  const int cpg = C / a.groups;
  const float* p1 = a.x1 + (long)b * a.HW * a.c1;
  const float* p2 = a.x2 ? a.x2 + (long)(a.bmod > 0 ? b % a.bmod : b) * a.HW * a.c2 : nullptr;
  const int px0 = slab * a.slab, px1 = min(a.HW, px0 + a.slab);
  const int t = threadIdx.x;
  int npl, slots, tq, tp;
  if (Q <= 256) { npl = 256 / Q; slots = 1; tq = t % Q; tp = t / Q; }
  else { npl = 1; slots = (Q + 255) / 256; tq = t; tp = 0; }
  float s4[GN_MAX_SLOTS][4], q4[GN_MAX_SLOTS][4];
#pragma unroll
  for (int k = 0; k < GN_MAX_SLOTS; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) { s4[k][j] = 0.f; q4[k][j] = 0.f; }
  if (tp < npl) {
    for (int pix = px0 + tp; pix < px1; pix += npl) {
#pragma unroll
      for (int k = 0; k < GN_MAX_SLOTS; ++k) {
        const int c = (tq + 256 * k) * 4;
        if (k < slots && c < C) {
          const float4 v = c < a.c1 ? *reinterpret_cast<const float4*>(p1 + (long)pix * a.c1 + c)
                                    : *reinterpret_cast<const float4*>(p2 + (long)pix * a.c2 + (c - a.c1));
          s4[k][0] += v.x; s4[k][1] += v.y; s4[k][2] += v.z; s4[k][3] += v.w;
          q4[k][0] += v.x * v.x; q4[k][1] += v.y * v.y; q4[k][2] += v.z * v.z; q4[k][3] += v.w * v.w;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < GN_MAX_SLOTS; ++k)
    if (k < slots) {
      float* d = part + ((long)k * 256 + t) * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) { d[j] = s4[k][j]; d[4 + j] = q4[k][j]; }
    }
  __syncthreads();
  if (t < a.groups) {
    double su = 0.0, sq = 0.0;
    for (int c = t * cpg; c < (t + 1) * cpg; ++c) {       // fixed order: channel, then pixel lane
      const int quad = c >> 2, j = c & 3;
      const int k = Q <= 256 ? 0 : quad / 256, tqq = Q <= 256 ? quad : quad % 256;
      for (int l = 0; l < npl; ++l) {
        const float* d = part + ((long)k * 256 + (l * (Q <= 256 ? Q : 0) + tqq)) * 8;
        su += (double)d[j];
        sq += (double)d[4 + j];
      }
    }
    a.stats[(((long)b * gridDim.y + slab) * a.groups + t) * 2] = su;
    a.stats[(((long)b * gridDim.y + slab) * a.groups + t) * 2 + 1] = sq;
  }
}

static int gn_slab_pixels(int C, int HW) {
  int slab = (16384 + C - 1) / C;   // ~64 KB of fp32 per block
  if (slab < 1) slab = 1;
  if (slab > HW) slab = HW;
  return slab;
}
extern "C" int stedm_gn_nslab(int C, int HW) {
  const int slab = gn_slab_pixels(C, HW);
  return (HW + slab - 1) / slab;
}

extern "C" int stedm_gn_stats(const float* x1, int c1, const float* x2, int c2, int x2_bmod, int groups, int B, int HW,
                              double* stats, void* stream) {
  STEDM_CHECK_ARG(x1 && stats, "gn_stats: null pointer");
  STEDM_CHECK_ARG((x2 != nullptr) == (c2 > 0), "gn_stats: x2/c2 mismatch");
  const int C = c1 + c2;
  STEDM_CHECK_ARG(groups > 0 && groups <= 64 && C % groups == 0 && C % 4 == 0 && c1 % 4 == 0 && C <= 1024 * GN_MAX_SLOTS,
                  "gn_stats: need groups <= 64, C %% groups == 0, C %% 4 == 0, c1 %% 4 == 0, C <= %d (C=%d c1=%d groups=%d)",
                  1024 * GN_MAX_SLOTS, C, c1, groups);
  const int slab = gn_slab_pixels(C, HW);
  GnStatsArgs a{x1, x2, c1, c2, x2_bmod, groups, HW, slab, stats};
  dim3 grid(B, (HW + slab - 1) / slab);
  const int slots = (C / 4 + 255) / 256;
  gn_stats_kernel<<<grid, 256, (size_t)slots * 256 * 8 * sizeof(float), as_stream(stream)>>>(a);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// (2) apply: elementwise over the virtual concat; 16 B in, 8 B (+8 B) out per lane, fully coalesced.
struct GnApplyArgs {
  const float* x1;
  const float* x2;
  int c1, c2, bmod, groups, HW, act;
  const float* gamma;
  const float* beta;
  float eps;
  const double* stats;
  void* out_hi;
  void* out_lo;
  long total_q;  // B*HW*C/4
  unsigned* ovf; // fp16 range guard flag (common.hpp) or nullptr
};

template <typename T>
__global__ void __launch_bounds__(256) gn_apply16_kernel(GnApplyArgs a, int slab, int nslab_stats) {
  typedef T V4 __attribute__((ext_vector_type(4)));
  __shared__ float lmean[64], lrstd[64];
  const int b = blockIdx.x, sl = blockIdx.y;
  const int C = a.c1 + a.c2, Q = C >> 2;
  const int cpg = C / a.groups;
  if (a.gamma && threadIdx.x < a.groups) {
    const double inv_n = 1.0 / ((double)cpg * a.HW);
    double su = 0.0, sq = 0.0;
    for (int k = 0; k < nslab_stats; ++k) {   // fixed order
      su += a.stats[(((long)b * nslab_stats + k) * a.groups + threadIdx.x) * 2];
      sq += a.stats[(((long)b * nslab_stats + k) * a.groups + threadIdx.x) * 2 + 1];
    }
    const double mean = su * inv_n;
    double var = sq * inv_n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    lmean[threadIdx.x] = (float)mean;
    lrstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)a.eps));
  }
  __syncthreads();
  const float* p1 = a.x1 + (long)b * a.HW * a.c1;
  const float* p2 = a.x2 ? a.x2 + (long)(a.bmod > 0 ? b % a.bmod : b) * a.HW * a.c2 : nullptr;
  const int px0 = sl * slab, px1 = min(a.HW, px0 + slab);
  const int total = (px1 - px0) * Q;
  int pix = px0 + threadIdx.x / Q, q = threadIdx.x % Q;
  const int dpix = 256 / Q, dq = 256 % Q;
  V4* oh = reinterpret_cast<V4*>(a.out_hi) + (long)b * a.HW * Q;
  V4* ol = a.out_lo ? reinterpret_cast<V4*>(a.out_lo) + (long)b * a.HW * Q : nullptr;
  unsigned bad = 0u;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int c = q * 4;
    float4 v = c < a.c1 ? *reinterpret_cast<const float4*>(p1 + (long)pix * a.c1 + c)
                        : *reinterpret_cast<const float4*>(p2 + (long)pix * a.c2 + (c - a.c1));
    if (a.gamma) {
      const float4 gm = *reinterpret_cast<const float4*>(a.gamma + c);
      const float4 bt = *reinterpret_cast<const float4*>(a.beta + c);
      if (cpg & 3) {
        const int g0 = c / cpg, g1 = (c + 1) / cpg, g2 = (c + 2) / cpg, g3 = (c + 3) / cpg;
        v.x = (v.x - lmean[g0]) * lrstd[g0] * gm.x + bt.x; v.y = (v.y - lmean[g1]) * lrstd[g1] * gm.y + bt.y;
        v.z = (v.z - lmean[g2]) * lrstd[g2] * gm.z + bt.z; v.w = (v.w - lmean[g3]) * lrstd[g3] * gm.w + bt.w;
      } else {
        const int g = c / cpg;
        const float mf = lmean[g], rstd = lrstd[g];
        v.x = (v.x - mf) * rstd * gm.x + bt.x; v.y = (v.y - mf) * rstd * gm.y + bt.y;
        v.z = (v.z - mf) * rstd * gm.z + bt.z; v.w = (v.w - mf) * rstd * gm.w + bt.w;
      }
    }
    if (a.act == 1) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
    V4 hi;
    hi[0] = (T)v.x; hi[1] = (T)v.y; hi[2] = (T)v.z; hi[3] = (T)v.w;
    const long o = (long)pix * Q + q;
    oh[o] = hi;
    bad |= f16_over_v4<T>(hi);
    if (ol) {
      V4 lo;
      lo[0] = (T)(v.x - (float)hi[0]); lo[1] = (T)(v.y - (float)hi[1]);
      lo[2] = (T)(v.z - (float)hi[2]); lo[3] = (T)(v.w - (float)hi[3]);
      ol[o] = lo;
    }
    pix += dpix; q += dq;
    if (q >= Q) { q -= Q; ++pix; }
  }
  f16_guard_commit(a.ovf, bad, a.gamma ? STEDM_F16G_NORM : STEDM_F16G_CAST);
}

extern "C" int stedm_gn_apply16(const float* x1, int c1, const float* x2, int c2, int x2_bmod, const float* gamma,
                                const float* beta, float eps, int groups, int act, const double* stats, int B, int HW,
                                void* out_hi, void* out_lo, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(x1 && out_hi, "gn_apply16: null pointer");
  STEDM_CHECK_ARG((x2 != nullptr) == (c2 > 0), "gn_apply16: x2/c2 mismatch");
  STEDM_CHECK_ARG((gamma == nullptr) || (beta && stats && groups > 0), "gn_apply16: gamma needs beta, stats, groups");
  const int C = c1 + c2;
  STEDM_CHECK_ARG(C % 4 == 0 && c1 % 4 == 0, "gn_apply16: channels must be multiples of 4");
  STEDM_CHECK_ARG(!gamma || C % groups == 0, "gn_apply16: C %% groups != 0");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "gn_apply16: bad mm_dtype");
  GnApplyArgs a{x1, x2, c1, c2, x2_bmod, groups > 0 ? groups : 1, HW, act, gamma, beta, eps, stats, out_hi, out_lo,
                (long)B * HW * (C / 4), mm_dtype == STEDM_F16 ? f16_guard_flag() : nullptr};
  STEDM_CHECK_ARG(!gamma || groups <= 64, "gn_apply16: groups <= 64");
  const int slab = gn_slab_pixels(C, HW);
  const int nslab = (HW + slab - 1) / slab;
  dim3 grid(B, nslab);
  if (mm_dtype == STEDM_F16)
    gn_apply16_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>(a, slab, nslab);
  else
    gn_apply16_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>(a, slab, nslab);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// ================================================================================================
// Producer-side statistics: per-(sample, 256-pixel slab, channel) partial sums
// ================================================================================================
// The convolution epilogues (conv_rs.inc) emit chan_stats[B][nslab][C][2] = {sum, sum of squares} of the tensor they
// write, one slab per 256-pixel M-tile; tensors from other producers get the same partials from gn_chan_stats_kernel. The
// apply kernel folds them into group statistics in a fixed order (bitwise reproducible), for ANY grouping of the virtual
// concat [x1 | x2] - the group boundaries of a decoder block straddle the concat seam - so one set of partials serves the
// next block's GroupNorm, the decoder's concat GroupNorm and nothing has to re-read the fp32 tensor for statistics.
extern "C" int stedm_gn_chan_nslab(int HW) { return (HW + 255) / 256; }

// grid (B, slots, blocks of qbs channel quads): 256 threads = QB quads x (256 / QB) pixel lanes
// CAST: the same read also leaves the plain 16-bit conversion of x (hi, and lo = x - hi when o_lo != NULL) — the training backward
// needs both of a gradient tensor (channel sums = bias gradient, 16-bit planes = operand of its dgrad / wgrad)
template <typename T, bool CAST>
__global__ void __launch_bounds__(256) gn_chan_stats_kernel(const float* __restrict__ x, int C, int HW, int slab_px, float* __restrict__ cs,
                                                            T* __restrict__ o_hi, T* __restrict__ o_lo, int qbs, unsigned* ovf = nullptr) {
  typedef T V4 __attribute__((ext_vector_type(4)));
  __shared__ float cpart[256 * 8];   // [npl][QB][8]
  const int b = blockIdx.x, slab = blockIdx.y, nslab = gridDim.y;
  const int Q = C >> 2, t = threadIdx.x;
  const int qb0 = blockIdx.z * qbs, QB = min(qbs, Q - qb0);   // qbs in {8, 16, 32, 64}: small tensors take narrow blocks, for a grid that fills the chip
  const int npl = 256 / QB, tq = t % QB, tp = t / QB;
  const float* px = x + (long)b * HW * C + (qb0 + tq) * 4;
  const int px0 = min(HW, slab * slab_px), px1 = min(HW, px0 + slab_px);   // trailing slots of an over-allocated partition stay 0
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  unsigned bad = 0u;
  if (tp < npl) {
#pragma unroll 4
    for (int pix = px0 + tp; pix < px1; pix += npl) {
      const float4 v = *reinterpret_cast<const float4*>(px + (long)pix * C);
      if constexpr (CAST) {
        const long o = ((long)b * HW + pix) * C + (qb0 + tq) * 4;
        V4 h4; h4[0] = (T)v.x; h4[1] = (T)v.y; h4[2] = (T)v.z; h4[3] = (T)v.w;
        *reinterpret_cast<V4*>(o_hi + o) = h4;
        bad |= f16_over_v4<T>(h4);
        if (o_lo) {
          V4 l4; l4[0] = (T)(v.x - (float)h4[0]); l4[1] = (T)(v.y - (float)h4[1]); l4[2] = (T)(v.z - (float)h4[2]); l4[3] = (T)(v.w - (float)h4[3]);
          *reinterpret_cast<V4*>(o_lo + o) = l4;
        }
      }
      s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
      q[0] += v.x * v.x; q[1] += v.y * v.y; q[2] += v.z * v.z; q[3] += v.w * v.w;
    }
    float* d = cpart + (tp * QB + tq) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) { d[j] = s[j]; d[4 + j] = q[j]; }
  }
  if constexpr (CAST) f16_guard_commit(ovf, bad, STEDM_F16G_CAST);
  __syncthreads();
  if (t < QB) {
    float su[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < npl; ++l) {   // fixed order
      const float* d = cpart + (l * QB + t) * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) { su[j] += d[j]; sq[j] += d[4 + j]; }
    }
    float* dst = cs + (((long)b * nslab + slab) * C + (qb0 + t) * 4) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) { dst[j * 2] = su[j]; dst[j * 2 + 1] = sq[j]; }
  }
}

// channel quads per block: 64 (1 KiB per pixel and block) unless the grid would leave most of the chip idle — a 16 x 16 x 512 tensor at
// batch 64 is 128 blocks of 64 quads; narrower blocks (>= 128 B per pixel) bring it to ~1024 (measured against fixed 64-quad blocks:
// training step 20.27 -> 20.17 ms).
static int chan_stats_qbs(int B, int nslab, int Q) {
  int qbs = 64;
  while (qbs > 8 && (long)B * nslab * ((Q + qbs - 1) / qbs) < 1024) qbs >>= 1;
  return qbs;
}

extern "C" int stedm_gn_chan_stats(const float* x, int C, int B, int HW, int nslab, float* chan_stats, void* stream) {
  STEDM_CHECK_ARG(x && chan_stats && C > 0 && C % 4 == 0 && B > 0 && HW > 0 && nslab >= 0, "gn_chan_stats: bad args (C %% 4)");
  const int Q = C / 4;
  if (nslab == 0) nslab = (HW + 255) / 256;
  const int slab_px = nslab == (HW + 255) / 256 ? 256 : (HW + nslab - 1) / nslab;   // the default partition is 256-pixel runs
  const int qbs = chan_stats_qbs(B, nslab, Q);
  dim3 grid(B, nslab, (Q + qbs - 1) / qbs);
  gn_chan_stats_kernel<__bf16, false><<<grid, 256, 0, as_stream(stream)>>>(x, C, HW, slab_px, chan_stats, nullptr, nullptr, qbs);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_gn_chan_stats16(const float* x, int C, int B, int HW, int nslab, float* chan_stats, void* out_hi, void* out_lo, int mm_dtype,
                                     void* stream) {
  STEDM_CHECK_ARG(x && chan_stats && out_hi && C > 0 && C % 4 == 0 && B > 0 && HW > 0 && nslab >= 0, "gn_chan_stats16: bad args (C %% 4)");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "gn_chan_stats16: bad mm_dtype");
  const int Q = C / 4;
  if (nslab == 0) nslab = (HW + 255) / 256;
  const int slab_px = nslab == (HW + 255) / 256 ? 256 : (HW + nslab - 1) / nslab;
  // every pixel must fall into a slot, or its 16-bit values would not be written
  STEDM_CHECK_ARG((long)nslab * slab_px >= HW, "gn_chan_stats16: the slots do not cover the sample");
  const int qbs = chan_stats_qbs(B, nslab, Q);
  dim3 grid(B, nslab, (Q + qbs - 1) / qbs);
  if (mm_dtype == STEDM_F16)
    gn_chan_stats_kernel<_Float16, true><<<grid, 256, 0, as_stream(stream)>>>(x, C, HW, slab_px, chan_stats, (_Float16*)out_hi, (_Float16*)out_lo, qbs,
                                                                                  f16_guard_flag());
  else
    gn_chan_stats_kernel<__bf16, true><<<grid, 256, 0, as_stream(stream)>>>(x, C, HW, slab_px, chan_stats, (__bf16*)out_hi, (__bf16*)out_lo, qbs);
  STEDM_LAUNCH_CHECK();
  return 0;
}

struct GnApplyCArgs {
  const float* x1;
  const float* x2;
  const float* cs1;
  const float* cs2;
  int c1, c2, bmod, groups, HW, act, nslab1, nslab2;
  const float* gamma;
  const float* beta;
  float eps;
  void* out_hi;
  void* out_lo;
  void* raw_hi;
  void* raw_lo;
  float* mr;     // optional [B][groups][2]: the {mean, rstd} this pass folds anyway, kept for the training backward
  unsigned* ovf; // fp16 range guard flag (common.hpp) or nullptr
  const void* x16;  // v8 kernel, X16 form: the x1 channels arrive as the 16-bit values a convolution's epilogue already wrote into channels
                    // [0, c1) of the raw plane [B][HW][c1 + c2] (= raw_hi): x1 is not read and that part of the raw plane not rewritten
};

// A value beyond the fp16 range (65 504) in a run of pixels makes that run's sum of squares exceed 65 504^2 = 4.29e9: a block whose channel
// partials all stay below the threshold cannot overflow and skips the per-element test of its stream (NaN / inf partials fail the `<` too).
#define STEDM_F16_SQ_SAFE 4.0e9f

// y = act(GroupNorm([x1|x2])) from channel partials -> 16-bit planes; optionally also the plain conversion of [x1|x2]
// (raw planes: the operand of the ResBlock's 1x1 skip convolution) from the same read of the fp32 tensors.
template <typename T>
__global__ void __launch_bounds__(256) gn_apply16c_kernel(GnApplyCArgs a, int slab) {
  typedef T V4 __attribute__((ext_vector_type(4)));
  __shared__ double dsu[256], dsq[256];
  __shared__ float lmean[64], lrstd[64];
  const int b = blockIdx.x, sl = blockIdx.y;
  const int C = a.c1 + a.c2, Q = C >> 2;
  const int cpg = C / a.groups;
  const int b2 = a.bmod > 0 ? b % a.bmod : b;
  const float* p1 = a.x1 + (long)b * a.HW * a.c1;
  const float* p2 = a.x2 ? a.x2 + (long)b2 * a.HW * a.c2 : nullptr;
  const int px0 = sl * slab, px1 = min(a.HW, px0 + slab);
  const int total = (px1 - px0) * Q;
  V4* oh = reinterpret_cast<V4*>(a.out_hi) + (long)b * a.HW * Q;
  V4* ol = a.out_lo ? reinterpret_cast<V4*>(a.out_lo) + (long)b * a.HW * Q : nullptr;
  V4* rh = a.raw_hi ? reinterpret_cast<V4*>(a.raw_hi) + (long)b * a.HW * Q : nullptr;
  V4* rl = a.raw_lo ? reinterpret_cast<V4*>(a.raw_lo) + (long)b * a.HW * Q : nullptr;
  // U independent cursors per thread (elements tid + k*256, advancing by U*256): the pass is a pure stream, its speed is the
  // number of 16-B loads in flight
  constexpr int U = 4;
  int pixs[U], qs[U];
#pragma unroll
  for (int k = 0; k < U; ++k) { const int e = threadIdx.x + k * 256; pixs[k] = px0 + e / Q; qs[k] = e % Q; }
  const int dpix = (256 * U) / Q, dq = (256 * U) % Q;
  // the first batch of loads is issued before the statistics prologue: its latency hides behind the partial-sum reduction
  float4 v[U];
  auto load_batch = [&](int i) {
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (i + k * 256 < total) {
        const int c = qs[k] * 4;
        v[k] = c < a.c1 ? *reinterpret_cast<const float4*>(p1 + (long)pixs[k] * a.c1 + c)
                        : *reinterpret_cast<const float4*>(p2 + (long)pixs[k] * a.c2 + (c - a.c1));
      }
    }
  };
  load_batch(threadIdx.x);
  bool maybe_over = false;       // block-uniform: some channel partial of this block's groups admits |x| > 65504 (fp16 guard, raw planes)
  bool norm_over = false;        // block-uniform: some channel's gamma / beta admits a normalised value beyond the fp16 range
  unsigned bad_raw = 0u, bad_norm = 0u;
  {
    const int L = 256 / a.groups;                 // lanes per group (groups <= 64)
    const int g = threadIdx.x / L, l = threadIdx.x % L;
    double su = 0.0, sq = 0.0;
    int big = 0;
    if (g < a.groups) {
      const int nmax = a.nslab1 > a.nslab2 ? a.nslab1 : a.nslab2;
      const int n = cpg * nmax;
      for (int e = l; e < n; e += L) {            // entry = (slab k, channel cc of the group); tensors may be partitioned differently
        const int k = e / cpg, c = g * cpg + (e - k * cpg);
        if (c < a.c1) {
          if (k < a.nslab1) { const float* p = a.cs1 + (((long)b * a.nslab1 + k) * a.c1 + c) * 2; su += (double)p[0]; sq += (double)p[1]; big |= !(p[1] < STEDM_F16_SQ_SAFE); }
        } else if (k < a.nslab2) {
          const float* p = a.cs2 + (((long)b2 * a.nslab2 + k) * a.c2 + (c - a.c1)) * 2; su += (double)p[0]; sq += (double)p[1]; big |= !(p[1] < STEDM_F16_SQ_SAFE);
        }
      }
    }
    dsu[threadIdx.x] = su; dsq[threadIdx.x] = sq;
    maybe_over = __syncthreads_or(big) != 0 && a.ovf != nullptr && __is_same(T, _Float16);
    if (g < a.groups && l == 0) {
      double s = 0.0, q = 0.0;
      for (int i = 0; i < L; ++i) { s += dsu[threadIdx.x + i]; q += dsq[threadIdx.x + i]; }   // fixed order
      const double inv_n = 1.0 / ((double)cpg * a.HW);
      const double mean = s * inv_n;
      double var = q * inv_n - mean * mean;
      var = var > 0.0 ? var : 0.0;
      lmean[g] = (float)mean;
      lrstd[g] = (float)(1.0 / sqrt(var + (double)a.eps));
      if (a.mr && sl == 0) { a.mr[((long)b * a.groups + g) * 2] = lmean[g]; a.mr[((long)b * a.groups + g) * 2 + 1] = lrstd[g]; }
    }
    // the normalised planes have their own switch: |x^| <= sqrt(n) whatever the raw values are, so a plane value can only leave the fp16
    // range through gamma / beta (sqrt(n) |gamma| + |beta| > 65504) - small raw values do not rule that out
    int nb = 0;
    if (a.ovf != nullptr && __is_same(T, _Float16)) {
      const float rn = sqrtf((float)cpg * (float)a.HW);
      for (int c = threadIdx.x; c < C; c += 256) nb |= !(rn * fabsf(a.gamma[c]) + fabsf(a.beta[c]) < 65504.f);
    }
    norm_over = __syncthreads_or(nb) != 0;
  }
  for (int i = threadIdx.x; i < total; i += 256 * U) {
    if (i != (int)threadIdx.x) load_batch(i);
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (i + k * 256 < total) {
        const int c = qs[k] * 4;
        float4 w = v[k];
        const long o = (long)pixs[k] * Q + qs[k];
        if (rh) {
          V4 hi;
          hi[0] = (T)w.x; hi[1] = (T)w.y; hi[2] = (T)w.z; hi[3] = (T)w.w;
          rh[o] = hi;
          if (maybe_over) bad_raw |= f16_over_v4<T>(hi);
          if (rl) {
            V4 lo;
            lo[0] = (T)(w.x - (float)hi[0]); lo[1] = (T)(w.y - (float)hi[1]);
            lo[2] = (T)(w.z - (float)hi[2]); lo[3] = (T)(w.w - (float)hi[3]);
            rl[o] = lo;
          }
        }
        const float4 gm = *reinterpret_cast<const float4*>(a.gamma + c);
        const float4 bt = *reinterpret_cast<const float4*>(a.beta + c);
        if (cpg & 3) {
          const int g0 = c / cpg, g1 = (c + 1) / cpg, g2 = (c + 2) / cpg, g3 = (c + 3) / cpg;
          w.x = (w.x - lmean[g0]) * lrstd[g0] * gm.x + bt.x; w.y = (w.y - lmean[g1]) * lrstd[g1] * gm.y + bt.y;
          w.z = (w.z - lmean[g2]) * lrstd[g2] * gm.z + bt.z; w.w = (w.w - lmean[g3]) * lrstd[g3] * gm.w + bt.w;
        } else {
          const int g = c / cpg;
          const float mf = lmean[g], rstd = lrstd[g];
          w.x = (w.x - mf) * rstd * gm.x + bt.x; w.y = (w.y - mf) * rstd * gm.y + bt.y;
          w.z = (w.z - mf) * rstd * gm.z + bt.z; w.w = (w.w - mf) * rstd * gm.w + bt.w;
        }
        if (a.act == 1) { w.x = silu_f(w.x); w.y = silu_f(w.y); w.z = silu_f(w.z); w.w = silu_f(w.w); }
        V4 hi;
        hi[0] = (T)w.x; hi[1] = (T)w.y; hi[2] = (T)w.z; hi[3] = (T)w.w;
        oh[o] = hi;
        if (norm_over) bad_norm |= f16_over_v4<T>(hi);
        if (ol) {
          V4 lo;
          lo[0] = (T)(w.x - (float)hi[0]); lo[1] = (T)(w.y - (float)hi[1]);
          lo[2] = (T)(w.z - (float)hi[2]); lo[3] = (T)(w.w - (float)hi[3]);
          ol[o] = lo;
        }
      }
      pixs[k] += dpix; qs[k] += dq;
      if (qs[k] >= Q) { qs[k] -= Q; ++pixs[k]; }
    }
  }
  if (maybe_over) f16_guard_commit(a.ovf, bad_raw, STEDM_F16G_RAW);
  if (norm_over) f16_guard_commit(a.ovf, bad_norm, STEDM_F16G_NORM);
}


// The same pass with 16-B accesses on BOTH sides: a lane still loads one coalesced float4 (4 channels) per cursor — a wave instruction
// reads 1 KiB contiguous — but works on TWO cursors at once (quads e and e + 256 of the block's run), and lane pairs swap halves before the
// store: the even lane ends up with the 8 contiguous channels of cursor 0, the odd lane with those of cursor 1, so every store is 16 B
// per lane (8-B stores run at 0.5-0.7 of the 16-B rate, MI355X_MICROARCH.md). The per-channel constants {mean, rstd * gamma, beta}
// come from LDS tables built once per block (no per-element group arithmetic, no gamma / beta loads in the stream).
// Needs C % 8 == 0, c1 % 8 == 0 and an even number of quads per block iteration (256 threads: always).
// cb > 0: the block takes a CHANNEL run of cb channels (whole groups, a multiple of 8) of every pixel of its sample instead of a pixel run
// of all channels: it folds the statistics and builds the table of its own groups only (the samples of the 8 x 8 level are 64 pixels of
// 1024 - 2048 channels: six pixel-run blocks per sample each built the whole table for 11 pixels of streaming).
template <typename T, bool X16 = false>
__global__ void __launch_bounds__(256) gn_apply16c_v8_kernel(GnApplyCArgs a, int slab, int cb) {
  typedef T V4 __attribute__((ext_vector_type(4)));
  __shared__ double dsu[256], dsq[256];
  __shared__ float lmean[64], lrstd[64];
  extern __shared__ __attribute__((aligned(16))) float tab[];   // [3][C]: mean, rstd * gamma, beta per channel
  const int b = blockIdx.x, sl = blockIdx.y;
  const int C = a.c1 + a.c2, Qall = C >> 2;
  const int cpg = C / a.groups;
  const int b2 = a.bmod > 0 ? b % a.bmod : b;
  const float* p1 = X16 ? nullptr : a.x1 + (long)b * a.HW * a.c1;
  const float* p2 = a.x2 ? a.x2 + (long)b2 * a.HW * a.c2 : nullptr;
  const T* p16 = X16 ? reinterpret_cast<const T*>(a.x16) + (long)b * a.HW * C : nullptr;      // rows of C channels: the raw plane itself
  const int c_lo = cb > 0 ? sl * cb : 0, c_n = cb > 0 ? cb : C;       // this block's channels
  const int Q = c_n >> 2, q_lo = c_lo >> 2;                           // quads per pixel of this block, first quad
  const int px0 = cb > 0 ? 0 : sl * slab, px1 = cb > 0 ? a.HW : min(a.HW, px0 + slab);
  const int total = (px1 - px0) * Q;             // quads of this block; even (Q is even)
  uint2* oh = reinterpret_cast<uint2*>(a.out_hi) + (long)b * a.HW * Qall;      // 8-B units (one quad of 16-bit values)
  uint2* ol = a.out_lo ? reinterpret_cast<uint2*>(a.out_lo) + (long)b * a.HW * Qall : nullptr;
  uint2* rh = a.raw_hi ? reinterpret_cast<uint2*>(a.raw_hi) + (long)b * a.HW * Qall : nullptr;
  uint2* rl = a.raw_lo ? reinterpret_cast<uint2*>(a.raw_lo) + (long)b * a.HW * Qall : nullptr;
  constexpr int U = 2;
  int pixs[U], qs[U];      // compute cursors
  int lpx[U], lqs[U];      // load cursors: run two block iterations ahead of the compute cursors
#pragma unroll
  for (int k = 0; k < U; ++k) { const int e = threadIdx.x + k * 256; pixs[k] = px0 + e / Q; qs[k] = e % Q; lpx[k] = pixs[k]; lqs[k] = qs[k]; }
  const int dpix = (256 * U) / Q, dq = (256 * U) % Q;
  // Three batches of loads in flight per thread (v: current, vn: next, vnn: the one after): the pass is a pure stream and its speed is the
  // bytes in flight. With one batch per thread (round 3) a CU's 3 blocks kept 24 KB in flight and nothing during the statistics prologue
  // (4.3 TB/s over a step's launches); now two batches (16 KB per block) are issued before the prologue and a third at the top of every
  // iteration. The loads are UNCONDITIONAL (an out-of-range cursor re-reads the block's first quad): with a branch around a load the compiler
  // cannot count vmcnt and waits for vmcnt(0) right after issuing the prefetch, which serialises the stream again; the full iterations of
  // the loop below are branch-free for the same reason, the (at most one) partial iteration is a predicated copy of the body.
  float4 v[U], vn[U], vnn[U];
  const int nfull = total / (256 * U);               // block iterations in which every lane of both cursors is live
  auto load_adv = [&](float4 (&dst)[U], int it) {    // loads block iteration `it` and advances the load cursors
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const bool ok = it * 256 * U + (int)threadIdx.x + k * 256 < total;
      const int px = ok ? lpx[k] : px0, c = (q_lo + (ok ? lqs[k] : 0)) * 4;
      if (X16 && c < a.c1) {           // 8 B of the producer's 16-bit values
        const V4 h4 = *reinterpret_cast<const V4*>(p16 + (long)px * C + c);
        dst[k] = make_float4((float)h4[0], (float)h4[1], (float)h4[2], (float)h4[3]);
      } else {
        const float* src = (!X16 && c < a.c1) ? p1 + (long)px * a.c1 + c : p2 + (long)px * a.c2 + (c - a.c1);
        dst[k] = *reinterpret_cast<const float4*>(src);
      }
      lpx[k] += dpix; lqs[k] += dq;
      if (lqs[k] >= Q) { lqs[k] -= Q; ++lpx[k]; }
    }
  };
  load_adv(v, 0);
  load_adv(vn, 1);
  bool maybe_over = false;       // block-uniform: some channel partial of this block's groups admits |x| > 65504 (fp16 guard, raw planes)
  bool norm_over = false;        // block-uniform: some channel's gamma / beta admits a normalised value beyond the fp16 range
  unsigned bad_raw = 0u, bad_norm = 0u;
  {
    const int g_lo = c_lo / cpg, ng = c_n / cpg;  // this block's groups
    const int L = 256 / ng;                       // lanes per group (groups <= 64)
    const int gl = threadIdx.x / L, l = threadIdx.x % L, g = g_lo + gl;
    double su = 0.0, sq = 0.0;
    int big = 0;
    if (gl < ng) {
      const int nmax = a.nslab1 > a.nslab2 ? a.nslab1 : a.nslab2;
      const int n = cpg * nmax;
      for (int e = l; e < n; e += L) {            // entry = (slab k, channel cc of the group); tensors may be partitioned differently
        const int k = e / cpg, c = g * cpg + (e - k * cpg);
        if (c < a.c1) {
          if (k < a.nslab1) { const float* p = a.cs1 + (((long)b * a.nslab1 + k) * a.c1 + c) * 2; su += (double)p[0]; sq += (double)p[1]; big |= !(p[1] < STEDM_F16_SQ_SAFE); }
        } else if (k < a.nslab2) {
          const float* p = a.cs2 + (((long)b2 * a.nslab2 + k) * a.c2 + (c - a.c1)) * 2; su += (double)p[0]; sq += (double)p[1]; big |= !(p[1] < STEDM_F16_SQ_SAFE);
        }
      }
    }
    dsu[threadIdx.x] = su; dsq[threadIdx.x] = sq;
    maybe_over = __syncthreads_or(big) != 0 && a.ovf != nullptr && __is_same(T, _Float16);
    if (gl < ng && l == 0) {
      double s = 0.0, q = 0.0;
      for (int i = 0; i < L; ++i) { s += dsu[threadIdx.x + i]; q += dsq[threadIdx.x + i]; }   // fixed order
      const double inv_n = 1.0 / ((double)cpg * a.HW);
      const double mean = s * inv_n;
      double var = q * inv_n - mean * mean;
      var = var > 0.0 ? var : 0.0;
      lmean[g] = (float)mean;
      lrstd[g] = (float)(1.0 / sqrt(var + (double)a.eps));
      // (pixel-run blocks all fold every group of their sample: the first writes; channel-run blocks own their groups)
      if (a.mr && (cb > 0 || sl == 0)) { a.mr[((long)b * a.groups + g) * 2] = lmean[g]; a.mr[((long)b * a.groups + g) * 2 + 1] = lrstd[g]; }
    }
    __syncthreads();
    int nb = 0;
    const bool guard = a.ovf != nullptr && __is_same(T, _Float16);
    const float rn = sqrtf((float)cpg * (float)a.HW);          // |x^| <= sqrt(n): the normalised planes' own switch (see gn_apply16c_kernel)
    for (int c = c_lo + threadIdx.x; c < c_lo + c_n; c += 256) {
      const int gc = c / cpg;
      const float gmc = a.gamma[c], btc = a.beta[c];
      tab[c] = lmean[gc];
      tab[C + c] = lrstd[gc] * gmc;
      tab[2 * C + c] = btc;
      if (guard) nb |= !(rn * fabsf(gmc) + fabsf(btc) < 65504.f);
    }
    norm_over = __syncthreads_or(nb) != 0;
  }
  const bool odd = threadIdx.x & 1;
  auto pack4 = [](float x0, float x1, float x2, float x3) {
    V4 h; h[0] = (T)x0; h[1] = (T)x1; h[2] = (T)x2; h[3] = (T)x3;
    return *reinterpret_cast<uint2*>(&h);
  };
  // lane pair exchange: the even lane keeps cursor 0's quad and receives its partner's; the odd lane keeps cursor 1's. Returns the
  // 16 B this lane stores: {even lane's quad, odd lane's quad} of the cursor it owns.
  auto pair16 = [&](uint2 q0, uint2 q1) {
    const uint2 give = odd ? q0 : q1;             // what the partner needs from me
    uint2 got;
    got.x = __shfl_xor(give.x, 1, 64); got.y = __shfl_xor(give.y, 1, 64);
    uint4 r;
    if (odd) { r.x = got.x; r.y = got.y; r.z = q1.x; r.w = q1.y; }      // cursor 1: even partner's quad first
    else { r.x = q0.x; r.y = q0.y; r.z = got.x; r.w = got.y; }
    return r;
  };
  auto body = [&](auto tail_c, const int it) {
    constexpr bool TAIL = decltype(tail_c)::value;
    load_adv(vnn, it + 2);
    // (total and 256 are even and the cursors advance together, so a lane pair is live or dead together for each cursor)
    const int i = it * 256 * U + (int)threadIdx.x;
    const bool live0 = !TAIL || i < total, live1 = !TAIL || i + 256 < total;
    uint2 oq[U], rq[U], olq[U], rlq[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int c = (q_lo + qs[k]) * 4;
      float4 w = v[k];
      const bool live = k == 0 ? live0 : live1;
      if (TAIL && !live) { w = make_float4(0.f, 0.f, 0.f, 0.f); }
      if (rh) {
        rq[k] = pack4(w.x, w.y, w.z, w.w);
        if (maybe_over) bad_raw |= f16_over4<T>(rq[k]);
        if (rl) {
          V4 hq = *reinterpret_cast<V4*>(&rq[k]);
          rlq[k] = pack4(w.x - (float)hq[0], w.y - (float)hq[1], w.z - (float)hq[2], w.w - (float)hq[3]);
        }
      }
      const int cc = live ? c : c_lo;
      const float4 mn = *reinterpret_cast<const float4*>(tab + cc);
      const float4 sc = *reinterpret_cast<const float4*>(tab + C + cc);
      const float4 bt = *reinterpret_cast<const float4*>(tab + 2 * C + cc);
      w.x = (w.x - mn.x) * sc.x + bt.x; w.y = (w.y - mn.y) * sc.y + bt.y; w.z = (w.z - mn.z) * sc.z + bt.z; w.w = (w.w - mn.w) * sc.w + bt.w;
      if (a.act == 1) { w.x = silu_f(w.x); w.y = silu_f(w.y); w.z = silu_f(w.z); w.w = silu_f(w.w); }
      oq[k] = pack4(w.x, w.y, w.z, w.w);
      if (norm_over) bad_norm |= f16_over4<T>(oq[k]);
      if (ol) {
        V4 hq = *reinterpret_cast<V4*>(&oq[k]);
        olq[k] = pack4(w.x - (float)hq[0], w.y - (float)hq[1], w.z - (float)hq[2], w.w - (float)hq[3]);
      }
    }
    // my cursor: 0 on even lanes, 1 on odd lanes; the 16-B destination starts at the EVEN lane's quad of that cursor
    const int mk = odd ? 1 : 0;
    const bool mlive = odd ? live1 : live0;
    const long o = (long)pixs[mk] * Qall + q_lo + (qs[mk] & ~1);
    const uint4 so = pair16(oq[0], oq[1]);
    if (!TAIL || mlive) *reinterpret_cast<uint4*>(oh + o) = so;
    if (ol) { const uint4 t4 = pair16(olq[0], olq[1]); if (!TAIL || mlive) *reinterpret_cast<uint4*>(ol + o) = t4; }
    // (X16: the x1 part of the raw plane is the source itself; c1 % 8 == 0 keeps a lane pair on one side of the seam)
    const bool rawst = !X16 || (q_lo + (qs[mk] & ~1)) * 4 >= a.c1;
    if (rh) { const uint4 t4 = pair16(rq[0], rq[1]); if ((!TAIL || mlive) && rawst) *reinterpret_cast<uint4*>(rh + o) = t4; }
    if (rl) { const uint4 t4 = pair16(rlq[0], rlq[1]); if ((!TAIL || mlive) && rawst) *reinterpret_cast<uint4*>(rl + o) = t4; }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      pixs[k] += dpix; qs[k] += dq;
      if (qs[k] >= Q) { qs[k] -= Q; ++pixs[k]; }
      v[k] = vn[k]; vn[k] = vnn[k];
    }
  };
  for (int it = 0; it < nfull; ++it) body(std::false_type{}, it);
  if (nfull * 256 * U < total) body(std::true_type{}, nfull);
  if (maybe_over) f16_guard_commit(a.ovf, bad_raw, STEDM_F16G_RAW);
  if (norm_over) f16_guard_commit(a.ovf, bad_norm, STEDM_F16G_NORM);
}

// Round 5: the pass on OCTETS (8 channels per thread and iteration), single-product modes. A lane's unit is 8 contiguous channels of one
// pixel: 16 B of 16-bit source (the h half of a decoder concat in its X16 form) or 32 B of fp32 source in, ONE 16-B store of the normalised
// plane out (plus one of the raw plane for fp32-source channels when asked) - no lane exchange, half the table reads and address
// arithmetic per byte of gn_apply16c_v8_kernel. (The quad kernel with 8-B loads for the 16-bit half ran the nine decoder concats at
// 3.1 TB/s of 1.64 GB where the fp32 form had run 5.2 TB/s of 2.41 GB: the pass is bound by requests in flight, not by bytes.)
// Same block partition as the quad kernel (pixel runs of all channels, or channel runs of whole groups: cb), same statistics fold.
// PAIR (C and c1 multiples of 16): the fp32 half is loaded with a 16-B lane stride - lanes 2k / 2k + 1 own octets e / e + 1 = quads 4 of one
// 64-B run; load 0 takes quads (0, 1) of the run, load 1 quads (2, 3), and the pair trades one quad each way (4 DPP moves) - instead of two
// 16-B loads at a 32-B lane stride, which request every cache line twice (3.4 TB/s on that half against 5.2 for coalesced loads).
template <typename T, bool X16, bool PAIR>
__global__ void __launch_bounds__(256) gn_apply16c_o8_kernel(GnApplyCArgs a, int slab, int cb) {
  typedef T V8 __attribute__((ext_vector_type(8)));
  __shared__ double dsu[256], dsq[256];
  __shared__ float lmean[64], lrstd[64];
  extern __shared__ __attribute__((aligned(16))) float tab[];   // [3][C]: mean, rstd * gamma, beta per channel
  const int b = blockIdx.x, sl = blockIdx.y;
  const int C = a.c1 + a.c2, Oall = C >> 3;
  const int cpg = C / a.groups;
  const int b2 = a.bmod > 0 ? b % a.bmod : b;
  const float* p1 = X16 ? nullptr : a.x1 + (long)b * a.HW * a.c1;
  const float* p2 = a.x2 ? a.x2 + (long)b2 * a.HW * a.c2 : nullptr;
  const T* p16 = X16 ? reinterpret_cast<const T*>(a.x16) + (long)b * a.HW * C : nullptr;      // rows of C channels: the raw plane itself
  const int c_lo = cb > 0 ? sl * cb : 0, c_n = cb > 0 ? cb : C;       // this block's channels
  const int O = c_n >> 3, o_lo = c_lo >> 3;                           // octets per pixel of this block, first octet
  const int px0 = cb > 0 ? 0 : sl * slab, px1 = cb > 0 ? a.HW : min(a.HW, px0 + slab);
  const int total = (px1 - px0) * O;             // octets of this block
  uint4* oh = reinterpret_cast<uint4*>(a.out_hi) + (long)b * a.HW * Oall;      // 16-B units (one octet of 16-bit values)
  uint4* rh = a.raw_hi ? reinterpret_cast<uint4*>(a.raw_hi) + (long)b * a.HW * Oall : nullptr;
  int pix = px0 + (int)threadIdx.x / O, oc = (int)threadIdx.x % O;        // compute cursor
  int lpx = pix, loc = oc;                                                // load cursor: two block iterations ahead
  const int dpix = 256 / O, dq = 256 % O;
  // three batches of loads in flight per thread, all unconditional (see gn_apply16c_v8_kernel): an out-of-range cursor re-reads the block's
  // first octet; a 16-bit-source octet issues its one 16-B load twice (same address) instead of a branch around the second
  float4 va[2], vb[2], vc[2];
  const int nfull = total / 256;
  auto load_adv = [&](float4 (&dst)[2], int it) {
    const bool ok = it * 256 + (int)threadIdx.x < total;
    const int px = ok ? lpx : px0, c = (o_lo + (ok ? loc : 0)) * 8;
    // one address pair, selected without a branch: the loads themselves are unconditional
    const bool h16 = X16 && c < a.c1;
    const float* f32 = (!X16 && c < a.c1) ? p1 + (long)px * a.c1 + c : p2 + (long)px * a.c2 + (c - a.c1);
    // (PAIR: the even lane of a pair starts at its own octet, the odd lane 16 B before its own: together 32 contiguous bytes per load)
    const char* s0 = h16 ? reinterpret_cast<const char*>(p16 + (long)px * C + c)
                         : reinterpret_cast<const char*>(f32) - ((PAIR && ok && (threadIdx.x & 1)) ? 16 : 0);
    const char* s1 = h16 ? s0 : s0 + ((PAIR && ok) ? 32 : 16);      // (a dead cursor re-reads the block's first octet: never before the tensor)
    dst[0] = *reinterpret_cast<const float4*>(s0);
    dst[1] = *reinterpret_cast<const float4*>(s1);
    lpx += dpix; loc += dq;
    if (loc >= O) { loc -= O; ++lpx; }
  };
  load_adv(va, 0);
  load_adv(vb, 1);
  bool maybe_over = false, norm_over = false;       // block-uniform fp16 guard switches (raw planes / normalised planes), as in the quad kernel
  unsigned bad_raw = 0u, bad_norm = 0u;
  {
    const int g_lo = c_lo / cpg, ng = c_n / cpg;  // this block's groups
    const int L = 256 / ng;                       // lanes per group (groups <= 64)
    const int gl = threadIdx.x / L, l = threadIdx.x % L, g = g_lo + gl;
    double su = 0.0, sq = 0.0;
    int big = 0;
    if (gl < ng) {
      const int nmax = a.nslab1 > a.nslab2 ? a.nslab1 : a.nslab2;
      const int n = cpg * nmax;
      for (int e = l; e < n; e += L) {            // entry = (slab k, channel cc of the group); tensors may be partitioned differently
        const int k = e / cpg, c = g * cpg + (e - k * cpg);
        if (c < a.c1) {
          if (k < a.nslab1) { const float* p = a.cs1 + (((long)b * a.nslab1 + k) * a.c1 + c) * 2; su += (double)p[0]; sq += (double)p[1]; big |= !(p[1] < STEDM_F16_SQ_SAFE); }
        } else if (k < a.nslab2) {
          const float* p = a.cs2 + (((long)b2 * a.nslab2 + k) * a.c2 + (c - a.c1)) * 2; su += (double)p[0]; sq += (double)p[1]; big |= !(p[1] < STEDM_F16_SQ_SAFE);
        }
      }
    }
    dsu[threadIdx.x] = su; dsq[threadIdx.x] = sq;
    maybe_over = __syncthreads_or(big) != 0 && a.ovf != nullptr && __is_same(T, _Float16);
    if (gl < ng && l == 0) {
      double s_ = 0.0, q_ = 0.0;
      for (int i = 0; i < L; ++i) { s_ += dsu[threadIdx.x + i]; q_ += dsq[threadIdx.x + i]; }   // fixed order
      const double inv_n = 1.0 / ((double)cpg * a.HW);
      const double mean = s_ * inv_n;
      double var = q_ * inv_n - mean * mean;
      var = var > 0.0 ? var : 0.0;
      lmean[g] = (float)mean;
      lrstd[g] = (float)(1.0 / sqrt(var + (double)a.eps));
      if (a.mr && (cb > 0 || sl == 0)) { a.mr[((long)b * a.groups + g) * 2] = lmean[g]; a.mr[((long)b * a.groups + g) * 2 + 1] = lrstd[g]; }
    }
    __syncthreads();
    int nb = 0;
    const bool guard = a.ovf != nullptr && __is_same(T, _Float16);
    const float rn = sqrtf((float)cpg * (float)a.HW);
    for (int c = c_lo + threadIdx.x; c < c_lo + c_n; c += 256) {
      const int gc = c / cpg;
      const float gmc = a.gamma[c], btc = a.beta[c];
      tab[c] = lmean[gc];
      tab[C + c] = lrstd[gc] * gmc;
      tab[2 * C + c] = btc;
      if (guard) nb |= !(rn * fabsf(gmc) + fabsf(btc) < 65504.f);
    }
    norm_over = __syncthreads_or(nb) != 0;
  }
  auto pack8 = [](const float (&x)[8]) {
    V8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (T)x[j];
    return __builtin_bit_cast(uint4, h);
  };
  auto over8 = [](uint4 q) { return f16_over2<T>(q.x) | f16_over2<T>(q.y) | f16_over2<T>(q.z) | f16_over2<T>(q.w); };
  auto body = [&](auto tail_c, const int it) {
    constexpr bool TAIL = decltype(tail_c)::value;
    load_adv(vc, it + 2);
    const bool live = !TAIL || it * 256 + (int)threadIdx.x < total;
    const int c = (o_lo + (live ? oc : 0)) * 8;
    float w[8];
    const bool from16 = X16 && c < a.c1;
    if (from16) {
      const V8 h8 = __builtin_bit_cast(V8, va[0]);
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = (float)h8[j];
    } else if (PAIR) {
      // even lane holds quads {0, 2} of the pair's 64-B run and owns {0, 1}; odd lane holds {1, 3} and owns {2, 3}
      const bool odd = threadIdx.x & 1;
      const float4 give = odd ? va[0] : va[1];
      float4 got;
      got.x = __shfl_xor(give.x, 1, 64); got.y = __shfl_xor(give.y, 1, 64); got.z = __shfl_xor(give.z, 1, 64); got.w = __shfl_xor(give.w, 1, 64);
      const float4 q0 = odd ? got : va[0], q1 = odd ? va[1] : got;
      w[0] = q0.x; w[1] = q0.y; w[2] = q0.z; w[3] = q0.w; w[4] = q1.x; w[5] = q1.y; w[6] = q1.z; w[7] = q1.w;
    } else {
      w[0] = va[0].x; w[1] = va[0].y; w[2] = va[0].z; w[3] = va[0].w; w[4] = va[1].x; w[5] = va[1].y; w[6] = va[1].z; w[7] = va[1].w;
    }
    const long o = (long)pix * Oall + o_lo + oc;
    if (rh && !from16) {          // (X16: the 16-bit half of the raw plane is the source itself)
      const uint4 rq = pack8(w);
      if (maybe_over) bad_raw |= over8(rq);
      if (live) rh[o] = rq;
    }
    const float4 m0 = *reinterpret_cast<const float4*>(tab + c), m1 = *reinterpret_cast<const float4*>(tab + c + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(tab + C + c), s1 = *reinterpret_cast<const float4*>(tab + C + c + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(tab + 2 * C + c), b1 = *reinterpret_cast<const float4*>(tab + 2 * C + c + 4);
    w[0] = (w[0] - m0.x) * s0.x + b0.x; w[1] = (w[1] - m0.y) * s0.y + b0.y; w[2] = (w[2] - m0.z) * s0.z + b0.z; w[3] = (w[3] - m0.w) * s0.w + b0.w;
    w[4] = (w[4] - m1.x) * s1.x + b1.x; w[5] = (w[5] - m1.y) * s1.y + b1.y; w[6] = (w[6] - m1.z) * s1.z + b1.z; w[7] = (w[7] - m1.w) * s1.w + b1.w;
    if (a.act == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = silu_f(w[j]);
    }
    const uint4 oq = pack8(w);
    if (norm_over) bad_norm |= over8(oq);
    if (live) oh[o] = oq;
    pix += dpix; oc += dq;
    if (oc >= O) { oc -= O; ++pix; }
    va[0] = vb[0]; va[1] = vb[1]; vb[0] = vc[0]; vb[1] = vc[1];
  };
  for (int it = 0; it < nfull; ++it) body(std::false_type{}, it);
  if (nfull * 256 < total) body(std::true_type{}, nfull);
  if (maybe_over) f16_guard_commit(a.ovf, bad_raw, STEDM_F16G_RAW);
  if (norm_over) f16_guard_commit(a.ovf, bad_norm, STEDM_F16G_NORM);
}

extern "C" int stedm_gn_apply16c_mr(const float* x1, int c1, const float* cs1, int nslab1, const float* x2, int c2, const float* cs2, int nslab2, int x2_bmod,
                                    const float* gamma, const float* beta, float eps, int groups, int act, int B, int HW,
                                    void* out_hi, void* out_lo, void* raw_hi, void* raw_lo, float* mean_rstd, int mm_dtype, void* stream);

static int gn_apply16c_launch(GnApplyCArgs a, int B, int mm_dtype, void* stream);

// The same pass when the x1 half of the concat already sits in the raw plane as 16-bit values (written there by the producing convolution's
// epilogue: stedm_conv_args.out16_hi + out16_stride, no fp32 tensor of it exists): reads 2 B instead of 4 for those channels and writes
// their normalised plane only. raw_hi [B][HW][c1 + c2] holds x1 in channels [0, c1) on entry and receives the plain conversion of x2 in
// [c1, c1 + c2). cs1: the channel statistics the producer's epilogue left (of the fp32 values before their rounding). Single-product modes.
extern "C" int stedm_gn_apply16c_x16(int c1, const float* cs1, int nslab1, const float* x2, int c2, const float* cs2, int nslab2, int x2_bmod,
                                     const float* gamma, const float* beta, float eps, int groups, int act, int B, int HW,
                                     void* out_hi, void* raw_hi, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(cs1 && out_hi && raw_hi && gamma && beta, "gn_apply16c_x16: null pointer");
  STEDM_CHECK_ARG((x2 != nullptr) == (c2 > 0) && (x2 == nullptr || cs2 != nullptr), "gn_apply16c_x16: x2/c2/cs2 mismatch");
  const int C = c1 + c2;
  STEDM_CHECK_ARG(C % 8 == 0 && c1 % 8 == 0 && c1 > 0 && (size_t)3 * C * sizeof(float) <= 48 * 1024, "gn_apply16c_x16: need c1, c1 + c2 multiples of 8, C <= 4096");
  STEDM_CHECK_ARG(groups > 0 && groups <= 64 && C % groups == 0, "gn_apply16c_x16: need groups <= 64 and C %% groups == 0");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "gn_apply16c_x16: bad mm_dtype");
  STEDM_CHECK_ARG(nslab1 > 0 && (x2 == nullptr || nslab2 > 0), "gn_apply16c_x16: nslab1 / nslab2 must be the slot counts of cs1 / cs2");
  GnApplyCArgs a{nullptr, x2, cs1, cs2, c1, c2, x2_bmod, groups, HW, act, nslab1, x2 ? nslab2 : 0, gamma, beta, eps, out_hi, nullptr, raw_hi, nullptr, nullptr,
                 mm_dtype == STEDM_F16 ? f16_guard_flag() : nullptr, raw_hi};
  return gn_apply16c_launch(a, B, mm_dtype, stream);
}

extern "C" int stedm_gn_apply16c(const float* x1, int c1, const float* cs1, int nslab1, const float* x2, int c2, const float* cs2, int nslab2, int x2_bmod,
                                 const float* gamma, const float* beta, float eps, int groups, int act, int B, int HW,
                                 void* out_hi, void* out_lo, void* raw_hi, void* raw_lo, int mm_dtype, void* stream) {
  return stedm_gn_apply16c_mr(x1, c1, cs1, nslab1, x2, c2, cs2, nslab2, x2_bmod, gamma, beta, eps, groups, act, B, HW, out_hi, out_lo, raw_hi, raw_lo,
                              nullptr, mm_dtype, stream);
}

extern "C" int stedm_gn_apply16c_mr(const float* x1, int c1, const float* cs1, int nslab1, const float* x2, int c2, const float* cs2, int nslab2, int x2_bmod,
                                    const float* gamma, const float* beta, float eps, int groups, int act, int B, int HW,
                                    void* out_hi, void* out_lo, void* raw_hi, void* raw_lo, float* mean_rstd, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(x1 && cs1 && out_hi && gamma && beta, "gn_apply16c: null pointer");
  STEDM_CHECK_ARG((x2 != nullptr) == (c2 > 0) && (x2 == nullptr || cs2 != nullptr), "gn_apply16c: x2/c2/cs2 mismatch");
  const int C = c1 + c2;
  STEDM_CHECK_ARG(C % 4 == 0 && c1 % 4 == 0 && C / 4 <= 1024, "gn_apply16c: channels must be multiples of 4, C <= 4096");
  STEDM_CHECK_ARG(groups > 0 && groups <= 64 && C % groups == 0, "gn_apply16c: need groups <= 64 and C %% groups == 0");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "gn_apply16c: bad mm_dtype");
  STEDM_CHECK_ARG(raw_hi || !raw_lo, "gn_apply16c: raw_lo without raw_hi");
  STEDM_CHECK_ARG(nslab1 > 0 && (x2 == nullptr || nslab2 > 0), "gn_apply16c: nslab1 / nslab2 must be the slot counts of cs1 / cs2");
  GnApplyCArgs a{x1, x2, cs1, cs2, c1, c2, x2_bmod, groups, HW, act, nslab1, x2 ? nslab2 : 0, gamma, beta, eps, out_hi, out_lo, raw_hi, raw_lo, mean_rstd,
                 mm_dtype == STEDM_F16 ? f16_guard_flag() : nullptr, nullptr};
  return gn_apply16c_launch(a, B, mm_dtype, stream);
}

static int gn_apply16c_launch(GnApplyCArgs a, int B, int mm_dtype, void* stream) {
  const int C = a.c1 + a.c2, HW = a.HW, groups = a.groups, c1 = a.c1;
  // Pixels per block. Every block folds the group statistics and builds its per-channel table first, a cost that grows with C, so wide
  // tensors want long runs (128 KiB of fp32 input: 21 pixels of the 1536-channel decoder concat; at a fixed 32 KiB that shape ran at 4.3
  // TB/s, now 5.3) — as long as the grid keeps >= 768 blocks (3 per CU), which the 64-pixel samples of the 8x8 level need (12 pixels per
  // block there: 3.1 -> 3.9 TB/s); narrow tensors are indifferent (tools/bench_gn.py). Uniform run lengths of 96 / 160 / 256 KiB cost the
  // denoising step 1.5 / 3 / 7 % (round 3)
  int slab;
  {
    const int want = (128 * 256 + C - 1) / C;                                   // 128 KiB of fp32 input
    int per_sample = (HW + want - 1) / want;                                    // blocks per sample at that run length
    const int need = (768 + B - 1) / B;                                         // ... and for 768 blocks in all
    if (per_sample < need) per_sample = need;
    if (per_sample > HW) per_sample = HW;
    slab = (HW + per_sample - 1) / per_sample;
  }
  if (slab < 1) slab = 1;
  if (slab > HW) slab = HW;
  dim3 grid(B, (HW + slab - 1) / slab);
  if (C % 8 == 0 && c1 % 8 == 0 && (size_t)3 * C * sizeof(float) <= 48 * 1024) {
    const size_t lds = (size_t)3 * C * sizeof(float);
    // Small samples cut by pixels would need several blocks per sample, each building the whole table: cut those by CHANNELS instead
    // (runs of whole groups, >= 512 B per pixel, on one side of the concat seam), so a block builds the table of its own groups only.
    int cb = 0;
    if (grid.y > 1 && HW <= 256) {
      const int cpg = C / groups;
      int unit = cpg;
      while (unit % 8 != 0) unit *= 2;
      for (int t = unit; t <= C / 2; t += unit)          // the widest run that still gives the grid its 768 blocks
        if (C % t == 0 && c1 % t == 0 && t >= 128 && (long)B * (C / t) >= 768) cb = t;
      if (cb > 0 && (C % cb != 0 || c1 % cb != 0 || cb % unit != 0 || cb / cpg > 64)) cb = 0;
    }
    if (cb > 0) grid = dim3(B, C / cb);
    static const int o8_mode = getenv("STEDM_GN_O8") ? atoi(getenv("STEDM_GN_O8")) : 1;      // A/B timing only: 0 quad kernel everywhere, 1 octets for the X16 form, 2 octets wherever they apply
    const bool o8_ok = !a.out_lo && !a.raw_lo && (cb == 0 || cb % 8 == 0);
    if (o8_ok && ((a.x16 && o8_mode >= 1) || o8_mode >= 2)) {
      static const bool no_pair = getenv("STEDM_GN_NOPAIR") != nullptr;      // A/B timing only
      const bool pair = C % 16 == 0 && c1 % 16 == 0 && (cb == 0 || cb % 16 == 0) && !no_pair;
#define GN_O8(TT, XX, PP) gn_apply16c_o8_kernel<TT, XX, PP><<<grid, 256, lds, as_stream(stream)>>>(a, slab, cb)
      if (mm_dtype == STEDM_F16) {
        if (a.x16) { if (pair) GN_O8(_Float16, true, true); else GN_O8(_Float16, true, false); }
        else { if (pair) GN_O8(_Float16, false, true); else GN_O8(_Float16, false, false); }
      } else {
        if (a.x16) { if (pair) GN_O8(__bf16, true, true); else GN_O8(__bf16, true, false); }
        else { if (pair) GN_O8(__bf16, false, true); else GN_O8(__bf16, false, false); }
      }
#undef GN_O8
      STEDM_LAUNCH_CHECK();
      return 0;
    }
    if (a.x16) {
      if (mm_dtype == STEDM_F16) gn_apply16c_v8_kernel<_Float16, true><<<grid, 256, lds, as_stream(stream)>>>(a, slab, cb);
      else gn_apply16c_v8_kernel<__bf16, true><<<grid, 256, lds, as_stream(stream)>>>(a, slab, cb);
    } else if (mm_dtype == STEDM_F16) gn_apply16c_v8_kernel<_Float16><<<grid, 256, lds, as_stream(stream)>>>(a, slab, cb);
    else gn_apply16c_v8_kernel<__bf16><<<grid, 256, lds, as_stream(stream)>>>(a, slab, cb);
    STEDM_LAUNCH_CHECK();
    return 0;
  }
  STEDM_CHECK_ARG(!a.x16, "gn_apply16c_x16: the 16-bit source form belongs to the vector kernel");
  if (mm_dtype == STEDM_F16)
    gn_apply16c_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>(a, slab);
  else
    gn_apply16c_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>(a, slab);
  STEDM_LAUNCH_CHECK();
  return 0;
}
