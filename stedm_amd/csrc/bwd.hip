// Backward-pass kernels of the training step (p_losses ddpm.py:1015-1048 under autograd, i.e. the reverse
// of UNetModel.forward openaimodel.py:761-806). The two heavy contractions of every convolution's backward run on the forward's
// own MFMA convolution kernels: dgrad = the same convolution with the flipped / transposed filter; wgrad = one GEMM
// dW[(tap, ci)][co] = sum_p col[(tap, ci)][p] * dY^T[co][p] over "transposed im2col" planes written here. Everything else
// (GroupNorm+SiLU backward, attention backward, reductions, loss, optimizer) is HBM-bound fp32 work in this file.
// No atomics: every reduction has a fixed order, gradients are bitwise reproducible.
#include "common.hpp"
#include "sgemm.hpp"

using namespace stedm;

namespace {

__device__ __forceinline__ float silu_grad(float y) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-y));
  return s * (1.0f + y * (1.0f - s));
}

// ------------------------------------------------------------------------------------------------ GroupNorm statistics
// chan partials of the (virtual concat) input -> mr[B][groups][2] = {mean, rstd}
__global__ void __launch_bounds__(256) gn_fold_kernel(const float* __restrict__ cs1, int nslab1, int c1, const float* __restrict__ cs2, int nslab2,
                                                      int c2, int groups, int HW, float eps, float* __restrict__ mr) {
  __shared__ double dsu[256], dsq[256];
  const int b = blockIdx.x, C = c1 + c2, cpg = C / groups;
  const int L = 256 / groups, g = threadIdx.x / L, l = threadIdx.x % L;
  double su = 0.0, sq = 0.0;
  if (g < groups) {
    const int nmax = nslab1 > nslab2 ? nslab1 : nslab2;
    for (int e = l; e < cpg * nmax; e += L) {
      const int k = e / cpg, c = g * cpg + (e - k * cpg);
      if (c < c1) {
        if (k < nslab1) { const float* p = cs1 + (((long)b * nslab1 + k) * c1 + c) * 2; su += (double)p[0]; sq += (double)p[1]; }
      } else if (k < nslab2) {
        const float* p = cs2 + (((long)b * nslab2 + k) * c2 + (c - c1)) * 2; su += (double)p[0]; sq += (double)p[1];
      }
    }
  }
  dsu[threadIdx.x] = su; dsq[threadIdx.x] = sq;
  __syncthreads();
  if (threadIdx.x < groups) {
    double s = 0.0, q = 0.0;
    for (int i = 0; i < L; ++i) { s += dsu[threadIdx.x * L + i]; q += dsq[threadIdx.x * L + i]; }
    const double n = (double)cpg * HW, mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    mr[((long)b * groups + threadIdx.x) * 2] = (float)mean;
    mr[((long)b * groups + threadIdx.x) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

struct GnBwdArgs {
  const float* x1; const float* x2; int c1, c2;
  const float* mr; const float* gamma; const float* beta; const float* dA;   // dA: [B][HW][C] gradient w.r.t. act(GN(x))
  int groups, HW, act;
  float* part;        // [B][nslab][C][2] {sum dy, sum dy*xhat}
  const float* gm;    // [B][groups][2]   {mean_g(gamma*dy), mean_g(gamma*dy*xhat)}
  const float* add;   // optional [B][HW][C] added to dx (the residual / skip branch of the block)
  float* dx1; float* dx2; int acc1, acc2;
  void* o_hi; void* o_lo;   // optional 16-bit planes of dx (operand of the producer convolution's dgrad / wgrad)
};

// grid (B, kGnBwdSlab-pixel slabs, blocks of 64 channel quads). 64-pixel slabs: with 256 the grid of a 16x16 x 512-channel tensor at batch 64
// was 128 blocks (half the chip idle, 64 dependent pixel steps per thread)
constexpr int kGnBwdSlab = 64;
__global__ void __launch_bounds__(256) gn_bwd_stats_kernel(GnBwdArgs a) {
  __shared__ float cpart[256 * 8];
  const int b = blockIdx.x, slab = blockIdx.y, nslab = gridDim.y;
  const int C = a.c1 + a.c2, Q = C >> 2, t = threadIdx.x, cpg = C / a.groups;
  const int qb0 = blockIdx.z * 64, QB = min(64, Q - qb0);
  const int npl = 256 / QB, tq = t % QB, tp = t / QB;
  const int c = (qb0 + tq) * 4;
  const int px0 = min(a.HW, slab * kGnBwdSlab), px1 = min(a.HW, px0 + kGnBwdSlab);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (tp < npl) {
    float mean[4], rstd[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = (c + j) / cpg;
      mean[j] = a.mr[((long)b * a.groups + g) * 2]; rstd[j] = a.mr[((long)b * a.groups + g) * 2 + 1];
      ga[j] = a.gamma[c + j]; be[j] = a.beta[c + j];
    }
    const float* px = c < a.c1 ? a.x1 + (long)b * a.HW * a.c1 + c : a.x2 + (long)b * a.HW * a.c2 + (c - a.c1);
    const int ldx = c < a.c1 ? a.c1 : a.c2;
    const float* pd = a.dA + (long)b * a.HW * C + c;
    for (int pix = px0 + tp; pix < px1; pix += npl) {
      const float4 xv = *reinterpret_cast<const float4*>(px + (long)pix * ldx);
      const float4 dv = *reinterpret_cast<const float4*>(pd + (long)pix * C);
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (xs[j] - mean[j]) * rstd[j];
        const float dy = a.act ? ds[j] * silu_grad(xh * ga[j] + be[j]) : ds[j];
        s1[j] += dy; s2[j] += dy * xh;
      }
    }
    float* d = cpart + (tp * QB + tq) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) { d[j] = s1[j]; d[4 + j] = s2[j]; }
  }
  __syncthreads();
  if (t < QB) {
    float u[4] = {0.f, 0.f, 0.f, 0.f}, w[4] = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < npl; ++l) {
      const float* d = cpart + (l * QB + t) * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) { u[j] += d[j]; w[j] += d[4 + j]; }
    }
    float* dst = a.part + (((long)b * nslab + slab) * C + (qb0 + t) * 4) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) { dst[j * 2] = u[j]; dst[j * 2 + 1] = w[j]; }
  }
}

// grid B: slab partials -> bc[B][C][2] (per-sample channel sums) and gm[B][groups][2]
__global__ void __launch_bounds__(256) gn_bwd_fold_kernel(const float* __restrict__ part, int nslab, int C, int groups, int HW,
                                                          const float* __restrict__ gamma, float* __restrict__ bc, float* __restrict__ gm) {
  extern __shared__ float sm[];   // [C][2] gamma-weighted sums
  const int b = blockIdx.x, cpg = C / groups;
  for (int c = threadIdx.x; c < C; c += 256) {
    float u = 0.f, w = 0.f;
    for (int k = 0; k < nslab; ++k) { const float* p = part + (((long)b * nslab + k) * C + c) * 2; u += p[0]; w += p[1]; }
    bc[((long)b * C + c) * 2] = u; bc[((long)b * C + c) * 2 + 1] = w;
    sm[c * 2] = u * gamma[c]; sm[c * 2 + 1] = w * gamma[c];
  }
  __syncthreads();
  if (threadIdx.x < groups) {
    double u = 0.0, w = 0.0;
    for (int c = threadIdx.x * cpg; c < (threadIdx.x + 1) * cpg; ++c) { u += (double)sm[c * 2]; w += (double)sm[c * 2 + 1]; }
    const double n = (double)cpg * HW;
    gm[((long)b * groups + threadIdx.x) * 2] = (float)(u / n);
    gm[((long)b * groups + threadIdx.x) * 2 + 1] = (float)(w / n);
  }
}

// dgamma[c] (+)= sum_b bc[b][c][1]; dbeta[c] (+)= sum_b bc[b][c][0]; grid ceil(C/16), 256 threads = 16 channels x 16 sample lanes: the kernel
// is pure load latency, so a lane walks few samples (4 at batch 64, their loads in flight together) and the grid is 4x the 64-channel form's
__global__ void __launch_bounds__(256) gn_bwd_param_kernel(const float* __restrict__ bc, int B, int C, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           int accumulate) {
  __shared__ float red[16][16][2];
  const int cl = threadIdx.x & 15, bl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
  float u = 0.f, w = 0.f;
  if (c < C) {
    int b = bl;
    for (; b + 48 < B; b += 64) {
      float2 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float2*>(bc + ((long)(b + 16 * i) * C + c) * 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) { u += v[i].x; w += v[i].y; }
    }
    for (; b < B; b += 16) { const float2 v = *reinterpret_cast<const float2*>(bc + ((long)b * C + c) * 2); u += v.x; w += v.y; }
  }
  red[bl][cl][0] = u; red[bl][cl][1] = w;
  __syncthreads();
  if (bl == 0 && c < C) {
    float su = 0.f, sw = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) { su += red[l][cl][0]; sw += red[l][cl][1]; }   // fixed order
    dbeta[c] = (accumulate ? dbeta[c] : 0.f) + su;
    dgamma[c] = (accumulate ? dgamma[c] : 0.f) + sw;
  }
}

// dx = rstd * (gamma*dy - m1 - xhat*m2) (+ add); grid (B, ceil(HW*Q / 1024)), one float4 per thread x 4
template <typename T>
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(GnBwdArgs a) {
  typedef T V4 __attribute__((ext_vector_type(4)));
  const int b = blockIdx.x, C = a.c1 + a.c2, Q = C >> 2, cpg = C / a.groups;
  const long total = (long)a.HW * Q;
  for (int k = 0; k < 4; ++k) {
    const long e = ((long)blockIdx.y * 4 + k) * 256 + threadIdx.x;
    if (e >= total) return;
    const int pix = (int)(e / Q), c = (int)(e % Q) * 4;
    const bool first = c < a.c1;
    const float4 xv = first ? *reinterpret_cast<const float4*>(a.x1 + ((long)b * a.HW + pix) * a.c1 + c)
                            : *reinterpret_cast<const float4*>(a.x2 + ((long)b * a.HW + pix) * a.c2 + (c - a.c1));
    const long o = ((long)b * a.HW + pix) * C + c;
    const float4 dv = *reinterpret_cast<const float4*>(a.dA + o);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
    // per-channel constants: 16-B loads of gamma / beta; the group's {mean, rstd} and {m1, m2} once per quad when a quad never straddles
    // two groups (channels per group % 4 == 0: every shape of the U-Net) — 4 loads instead of 24 scalar ones per quad
    const float4 ga4 = *reinterpret_cast<const float4*>(a.gamma + c), be4 = *reinterpret_cast<const float4*>(a.beta + c);
    const float gas[4] = {ga4.x, ga4.y, ga4.z, ga4.w}, bes[4] = {be4.x, be4.y, be4.z, be4.w};
    float mean[4], rstd[4], m1[4], m2[4];
    if ((cpg & 3) == 0) {
      const long gi = ((long)b * a.groups + c / cpg) * 2;
      const float2 mr = *reinterpret_cast<const float2*>(a.mr + gi), gm = *reinterpret_cast<const float2*>(a.gm + gi);
#pragma unroll
      for (int j = 0; j < 4; ++j) { mean[j] = mr.x; rstd[j] = mr.y; m1[j] = gm.x; m2[j] = gm.y; }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long gi = ((long)b * a.groups + (c + j) / cpg) * 2;
        mean[j] = a.mr[gi]; rstd[j] = a.mr[gi + 1]; m1[j] = a.gm[gi]; m2[j] = a.gm[gi + 1];
      }
    }
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (xs[j] - mean[j]) * rstd[j];
      const float dy = a.act ? ds[j] * silu_grad(xh * gas[j] + bes[j]) : ds[j];
      r[j] = rstd[j] * (gas[j] * dy - m1[j] - xh * m2[j]);
    }
    if (a.add) {
      const float4 av = *reinterpret_cast<const float4*>(a.add + o);
      r[0] += av.x; r[1] += av.y; r[2] += av.z; r[3] += av.w;
    }
    float* dst = first ? a.dx1 + ((long)b * a.HW + pix) * a.c1 + c : a.dx2 + ((long)b * a.HW + pix) * a.c2 + (c - a.c1);
    if (first ? a.acc1 : a.acc2) {
      const float4 old = *reinterpret_cast<const float4*>(dst);
      r[0] += old.x; r[1] += old.y; r[2] += old.z; r[3] += old.w;
    }
    *reinterpret_cast<float4*>(dst) = make_float4(r[0], r[1], r[2], r[3]);
    if (a.o_hi) {
      V4 hi;
#pragma unroll
      for (int j = 0; j < 4; ++j) hi[j] = (T)r[j];
      reinterpret_cast<V4*>(a.o_hi)[o >> 2] = hi;
      if (a.o_lo) {
        V4 lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) lo[j] = (T)(r[j] - (float)hi[j]);
        reinterpret_cast<V4*>(a.o_lo)[o >> 2] = lo;
      }
    }
  }
}

// One-pass form: a block owns (sample b, a run of `cb` channels made of whole groups) and keeps xhat and dy of its HW x cb slice in REGISTERS
// (512 threads = qb channel quads x 512 / qb pixel lanes, PPT pixels per thread), so x and dA are read ONCE, with every load of the slice
// in flight at once, and the group means never leave the block: statistics, fold and apply in one launch (the three-kernel chain above
// cost ~50 us of launches and tails per GroupNorm on tensors that stream in ~25 us). Writes bc[b][c][2] for gn_bwd_param_kernel.
struct GnBwdFusedArgs {
  GnBwdArgs g;
  float* bc;
  int cb;
};

template <typename T, int PPT>
__global__ void __launch_bounds__(512) gn_bwd_fused_kernel(GnBwdFusedArgs fa) {
  typedef T V4 __attribute__((ext_vector_type(4)));
  __shared__ float red[512 * 8 + 2 * 288 + 2 * 72];  // [npl][qb][8] lane partials | [cb][2] gamma-weighted sums | [groups of the block][2]
  const GnBwdArgs& a = fa.g;
  const int b = blockIdx.x, cb = fa.cb, c0 = blockIdx.y * cb;
  const int C = a.c1 + a.c2, cpg = C / a.groups, qb = cb >> 2, t = threadIdx.x;
  const int npl = 512 / qb, tq = t % qb, tp = t / qb;
  const bool on = tp < npl;
  const int c = c0 + tq * 4;
  const bool first = c0 < a.c1;                      // the host keeps a block on one side of the concat seam
  const float* px = first ? a.x1 + (long)b * a.HW * a.c1 + c : a.x2 + (long)b * a.HW * a.c2 + (c - a.c1);
  const int ldx = first ? a.c1 : a.c2;
  const float* pd = a.dA + (long)b * a.HW * C + c;
  float mean[4], rstd[4], ga[4], be[4];
  float xh[PPT][4], dy[PPT][4];
  float4 av[PPT];
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (on) {
    float4 xv[PPT], dv[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int pix = tp + k * npl;
      xv[k] = dv[k] = av[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pix < a.HW) {
        xv[k] = *reinterpret_cast<const float4*>(px + (long)pix * ldx);
        dv[k] = *reinterpret_cast<const float4*>(pd + (long)pix * C);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = (c + j) / cpg;
      mean[j] = a.mr[((long)b * a.groups + g) * 2]; rstd[j] = a.mr[((long)b * a.groups + g) * 2 + 1];
      ga[j] = a.gamma[c + j]; be[j] = a.beta[c + j];
    }
    if (a.add) {     // the residual branch's gradient is needed after the fold: in flight across it
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const int pix = tp + k * npl;
        if (pix < a.HW) av[k] = *reinterpret_cast<const float4*>(a.add + ((long)b * a.HW + pix) * C + c);
      }
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const bool live = tp + k * npl < a.HW;
      const float xi[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w}, di[4] = {dv[k].x, dv[k].y, dv[k].z, dv[k].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xh[k][j] = (xi[j] - mean[j]) * rstd[j];
        dy[k][j] = a.act ? di[j] * silu_grad(xh[k][j] * ga[j] + be[j]) : di[j];
        if (live) { s1[j] += dy[k][j]; s2[j] += dy[k][j] * xh[k][j]; }
      }
    }
    float* d = red + (tp * qb + tq) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) { d[j] = s1[j]; d[4 + j] = s2[j]; }
  }
  __syncthreads();
  float* gsum = red + 512 * 8;                       // [cb][2]
  float* gmv = gsum + 2 * 288;                       // [cb / cpg][2]  (cb <= 288)
  if (t < qb) {
    float u[4] = {0.f, 0.f, 0.f, 0.f}, w[4] = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < npl; ++l) {                  // fixed order
      const float* d = red + (l * qb + t) * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) { u[j] += d[j]; w[j] += d[4 + j]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cc = c0 + t * 4 + j;
      fa.bc[((long)b * C + cc) * 2] = u[j]; fa.bc[((long)b * C + cc) * 2 + 1] = w[j];
      const float gmm = a.gamma[cc];
      gsum[(t * 4 + j) * 2] = u[j] * gmm; gsum[(t * 4 + j) * 2 + 1] = w[j] * gmm;
    }
  }
  __syncthreads();
  if (t < cb / cpg) {
    double u = 0.0, w = 0.0;
    for (int k = t * cpg; k < (t + 1) * cpg; ++k) { u += (double)gsum[k * 2]; w += (double)gsum[k * 2 + 1]; }
    const double n = (double)cpg * a.HW;
    gmv[t * 2] = (float)(u / n); gmv[t * 2 + 1] = (float)(w / n);
  }
  __syncthreads();
  if (!on) return;
  float m1[4], m2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int gl = (tq * 4 + j) / cpg; m1[j] = gmv[gl * 2]; m2[j] = gmv[gl * 2 + 1]; }
  float* pdx = first ? a.dx1 + (long)b * a.HW * a.c1 + c : a.dx2 + (long)b * a.HW * a.c2 + (c - a.c1);
  const bool acc = first ? a.acc1 : a.acc2;
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int pix = tp + k * npl;
    if (pix >= a.HW) break;
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = rstd[j] * (ga[j] * dy[k][j] - m1[j] - xh[k][j] * m2[j]);
    r[0] += av[k].x; r[1] += av[k].y; r[2] += av[k].z; r[3] += av[k].w;
    const long o = ((long)b * a.HW + pix) * C + c;
    float* dst = pdx + (long)pix * ldx;
    if (acc) {
      const float4 old = *reinterpret_cast<const float4*>(dst);
      r[0] += old.x; r[1] += old.y; r[2] += old.z; r[3] += old.w;
    }
    *reinterpret_cast<float4*>(dst) = make_float4(r[0], r[1], r[2], r[3]);
    if (a.o_hi) {
      V4 hi;
#pragma unroll
      for (int j = 0; j < 4; ++j) hi[j] = (T)r[j];
      reinterpret_cast<V4*>(a.o_hi)[o >> 2] = hi;
      if (a.o_lo) {
        V4 lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) lo[j] = (T)(r[j] - (float)hi[j]);
        reinterpret_cast<V4*>(a.o_lo)[o >> 2] = lo;
      }
    }
  }
}

template <typename T>
static void launch_gn_bwd_fused(const GnBwdFusedArgs& fa, int ppt, dim3 grid, hipStream_t st) {
  if (ppt <= 2) gn_bwd_fused_kernel<T, 2><<<grid, 512, 0, st>>>(fa);
  else if (ppt <= 4) gn_bwd_fused_kernel<T, 4><<<grid, 512, 0, st>>>(fa);
  else gn_bwd_fused_kernel<T, 8><<<grid, 512, 0, st>>>(fa);
}

// channel run of the one-pass kernel for this shape (0: use the three-kernel chain): whole groups, one side of the concat seam, rows of
// at least 64 B (128 B preferred), at most 8 pixels per thread; *ppt_out = pixels per thread
static int gn_bwd_fused_cb(int c1, int c2, int groups, int HW, int* ppt_out) {
  const int C = c1 + c2, cpg = C / groups;
  int unit = cpg;
  while (unit % 4 != 0) unit *= 2;                   // whole groups and whole quads
  auto ppt_of = [&](int cb) { const int npl = 512 / (cb / 4); return npl > 0 ? (HW + npl - 1) / npl : 1 << 20; };
  // the widest run at <= 4 pixels per thread (88 registers: two blocks per CU) with rows of >= 128 B; else the widest at <= 8 (164 registers)
  int best = 0;
  for (int pass = 0; pass < 2 && best == 0; ++pass)
    for (int cb = unit; cb <= 288; cb += unit) {
      if (c1 % cb != 0 || (c2 != 0 && c2 % cb != 0) || cb < (pass == 0 ? 32 : 16) || ppt_of(cb) > (pass == 0 ? 4 : 8)) continue;
      best = cb;
    }
  if (best == 0) return 0;
  *ppt_out = ppt_of(best);
  return best;
}

// ------------------------------------------------------------------------------------------------ transposed im2col (16-bit)
// src [B][Hs][Ws][C] 16-bit NHWC -> dst [(tap*C + c)][Ppad], p = (b*Ho + y)*Wo + x over the OUTPUT grid of the convolution;
// mode 0: stride 1 (Ho = Hs); 1: nearest-2x upsample then conv (Ho = 2Hs); 2: stride 2 (Ho = Hs/2). ks 1 or 3 (pad ks/2).
// grid (Ppad/64, ceil(C/64), taps), 256 threads: a 64-pixel x 64-channel tile goes through LDS.
__global__ void __launch_bounds__(256) im2col_t16_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int B, int Hs, int Ws, int C, int Ho,
                                                         int Wo, int ks, int mode, long P, long Ppad) {
  __shared__ uint16_t tile[64][72];
  const int tap = blockIdx.z, ky = tap / ks - ks / 2, kx = tap % ks - ks / 2;
  const long p0 = (long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int e = threadIdx.x + it * 256, pl = e >> 3, cq = (e & 7) * 8;
    const long p = p0 + pl;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (p < P && c0 + cq < C) {
      const int x = (int)(p % Wo), y = (int)((p / Wo) % Ho), b = (int)(p / ((long)Wo * Ho));
      int sy, sx; bool ok;
      if (mode == 1) { const int uy = y + ky, ux = x + kx; ok = uy >= 0 && uy < Ho && ux >= 0 && ux < Wo; sy = uy >> 1; sx = ux >> 1; }
      else if (mode == 2) { sy = 2 * y + ky; sx = 2 * x + kx; ok = sy >= 0 && sy < Hs && sx >= 0 && sx < Ws; }
      else { sy = y + ky; sx = x + kx; ok = sy >= 0 && sy < Hs && sx >= 0 && sx < Ws; }
      if (ok) v = *reinterpret_cast<const uint4*>(src + (((long)b * Hs + sy) * Ws + sx) * C + c0 + cq);
    }
    *reinterpret_cast<uint4*>(&tile[pl][cq]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int e = threadIdx.x + it * 256, cl = e >> 3, pq = (e & 7) * 8;
    if (c0 + cl >= C) continue;
    uint16_t r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = tile[pq + j][cl];
    uint4 v;
    v.x = r[0] | ((uint32_t)r[1] << 16); v.y = r[2] | ((uint32_t)r[3] << 16);
    v.z = r[4] | ((uint32_t)r[5] << 16); v.w = r[6] | ((uint32_t)r[7] << 16);
    *reinterpret_cast<uint4*>(dst + ((long)tap * C + c0 + cl) * Ppad + p0 + pq) = v;
  }
}

// dw [taps][cin_ld][cout_ld] fp32 (GEMM result) -> grad OIHW [cout][cin][taps] (+= when accumulate).
// grid (ceil(cout/32), ceil(cin/CIT)): a [taps][CIT ci][32 co] block goes through LDS: 128-B runs in, (CIT ci x taps)-float runs out.
// CIT = 16; 4 for small filters with many split-K partials (128 -> 128: a 4 x 8 grid of blocks walked 32 partials serially in 20 us).
template <int CIT>
__global__ void __launch_bounds__(256) wgrad_to_oihw_kernel(const float* __restrict__ dw, float* __restrict__ grad, int cout, int cin, int taps, int cin_ld,
                                                            int cout_ld, int accumulate, int nsplit) {
  __shared__ float tile[9 * CIT][33];
  constexpr int SH = CIT == 16 ? 4 : 2;
  const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * CIT;
  if ((cout_ld & 3) == 0 && co0 + 32 <= cout && (reinterpret_cast<uintptr_t>(dw) & 15) == 0) {
    // 16-B loads along co (8 lanes cover a 128-B run): a quarter of the load instructions, four times the bytes in flight per lane
    for (int e = threadIdx.x; e < taps * CIT * 8; e += 256) {
      const int c4 = e & 7, r = e >> 3;               // r = tap * CIT + ci_local
      const int tap = r >> SH, cil = r & (CIT - 1);
      const int ci = ci0 + cil;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ci < cin)
#pragma unroll 8
        for (int z = 0; z < nsplit; ++z) {            // split-K partials, fixed order
          const float4 w = *reinterpret_cast<const float4*>(dw + (((long)z * taps + tap) * cin_ld + ci) * cout_ld + co0 + c4 * 4);
          v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
      tile[r][c4 * 4] = v.x; tile[r][c4 * 4 + 1] = v.y; tile[r][c4 * 4 + 2] = v.z; tile[r][c4 * 4 + 3] = v.w;
    }
  } else
  for (int e = threadIdx.x; e < taps * CIT * 32; e += 256) {
    const int col = e & 31, r = e >> 5;               // r = tap * CIT + ci_local
    const int tap = r >> SH, cil = r & (CIT - 1);
    const int co = co0 + col, ci = ci0 + cil;
    float v = 0.f;
    if (co < cout && ci < cin)
#pragma unroll 4
      for (int z = 0; z < nsplit; ++z) v += dw[(((long)z * taps + tap) * cin_ld + ci) * cout_ld + co];    // split-K partials, fixed order
    tile[r][col] = v;
  }
  __syncthreads();
  const int run = CIT * taps;                         // contiguous floats of one output row: [ci0 .. ci0+CIT-1][taps]
  for (int e = threadIdx.x; e < 32 * run; e += 256) {
    const int col = e / run, k = e - col * run;       // k = ci_local * taps + tap
    const int cil = k / taps, tap = k - cil * taps;
    const int co = co0 + col, ci = ci0 + cil;
    if (co < cout && ci < cin) {
      const long o = ((long)co * cin + ci) * taps + tap;
      grad[o] = (accumulate ? grad[o] : 0.f) + tile[tap * CIT + cil][col];
    }
  }
}

// cs [B][nslab][C][2] (sums in [..][0]) -> per_sample[b*ld + c] (optional) and total[c] (+= when accumulate; optional)
// grid ceil(C/16); 256 threads = 16 channels x 16 sample lanes (pure load latency: few samples per lane, their loads in flight together;
// fixed-order fold of the 16 lanes through LDS)
__global__ void __launch_bounds__(256) chan_sum_fold_kernel(const float* __restrict__ cs, int B, int nslab, int C, float* __restrict__ per_sample, long ld,
                                                            float* __restrict__ total, int accumulate, float* __restrict__ total2) {
  __shared__ float red[16][16];
  const int cl = threadIdx.x & 15, bl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
  float tot = 0.f;
  if (c < C) {
    int b = bl;
    for (; b + 48 < B; b += 64) {
      float u[4] = {0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < nslab; ++k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] += cs[(((long)(b + 16 * i) * nslab + k) * C + c) * 2];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (per_sample) per_sample[(long)(b + 16 * i) * ld + c] = u[i];
        tot += u[i];
      }
    }
    for (; b < B; b += 16) {
      float u = 0.f;
      for (int k = 0; k < nslab; ++k) u += cs[(((long)b * nslab + k) * C + c) * 2];
      if (per_sample) per_sample[(long)b * ld + c] = u;
      tot += u;
    }
  }
  red[bl][cl] = tot;
  __syncthreads();
  if (bl == 0 && c < C && total) {
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) t += red[l][cl];   // fixed order
    const float v = (accumulate ? total[c] : 0.f) + t;
    total[c] = v;
    if (total2) total2[c] = v;      // a second parameter with the same gradient (two biases added onto one tensor)
  }
}

// ------------------------------------------------------------------------------------------------ resampling
// out[b][y][x][c] (+)= sum of the 2x2 block of in [B][2H][2W][C]  (backward of the nearest-neighbour 2x upsample)
__global__ void sum2x2_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int Q, int accumulate, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int q = (int)(i % Q), x = (int)((i / Q) % W), y = (int)((i / ((long)Q * W)) % H), b = (int)(i / ((long)Q * W * H));
  const float4* p = reinterpret_cast<const float4*>(in) + (((long)b * 2 * H + 2 * y) * 2 * W + 2 * x) * Q + q;
  const float4 a0 = p[0], a1 = p[Q], a2 = p[(long)2 * W * Q], a3 = p[(long)2 * W * Q + Q];
  float4 r = make_float4(a0.x + a1.x + a2.x + a3.x, a0.y + a1.y + a2.y + a3.y, a0.z + a1.z + a2.z + a3.z, a0.w + a1.w + a2.w + a3.w);
  float4* o = reinterpret_cast<float4*>(out) + i;
  if (accumulate) { const float4 v = *o; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
  *o = r;
}

// 16-bit planes [B][2Ho][2Wo][C]: value of in[b][y/2][x/2][c] at even (y, x), zero elsewhere (dgrad of a stride-2 convolution
// is the stride-1 convolution of this zero-inserted gradient with the flipped filter)
template <typename T>
__global__ void zero_insert16_kernel(const float* __restrict__ in, T* __restrict__ hi, T* __restrict__ lo, int Ho, int Wo, int Q, long total) {
  typedef T V4 __attribute__((ext_vector_type(4)));
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over the [B][2Ho][2Wo][Q] output
  if (i >= total) return;
  const int q = (int)(i % Q), x = (int)((i / Q) % (2 * Wo)), y = (int)((i / ((long)Q * 2 * Wo)) % (2 * Ho)), b = (int)(i / ((long)Q * 4 * Wo * Ho));
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!(x & 1) && !(y & 1)) v = reinterpret_cast<const float4*>(in)[(((long)b * Ho + (y >> 1)) * Wo + (x >> 1)) * Q + q];
  V4 h; h[0] = (T)v.x; h[1] = (T)v.y; h[2] = (T)v.z; h[3] = (T)v.w;
  reinterpret_cast<V4*>(hi)[i] = h;
  if (lo) {
    V4 l; l[0] = (T)(v.x - (float)h[0]); l[1] = (T)(v.y - (float)h[1]); l[2] = (T)(v.z - (float)h[2]); l[3] = (T)(v.w - (float)h[3]);
    reinterpret_cast<V4*>(lo)[i] = l;
  }
}

// ------------------------------------------------------------------------------------------------ attention backward
// QKVAttentionLegacy (openaimodel.py:378-394) reversed; one block per (sample, head); P and dS live in LDS ([T][T] fp32 each), the
// q / k / v / dO operands are staged through LDS 32 channels at a time ([T][33] images).
__global__ void __launch_bounds__(256) attn_legacy_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dO, float* __restrict__ dqkv, int T,
                                                              int heads, int ch) {
  extern __shared__ float sm[];
  float* P = sm;
  float* dS = sm + (long)T * T;
  float* st = dS + (long)T * T;            // 4 staging images [T][33]
  float* sq = st; float* sk = st + T * 33; float* sv = st + 2 * T * 33; float* sg = st + 3 * T * 33;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const long ld = (long)heads * 3 * ch, ldo = (long)heads * ch;
  const float* q = qkv + (long)b * T * ld + (long)h * 3 * ch;
  const float* k = q + ch;
  const float* v = q + 2 * ch;
  const float* go = dO + (long)b * T * ldo + (long)h * ch;
  float* dq = dqkv + (long)b * T * ld + (long)h * 3 * ch;
  float* dk = dq + ch;
  float* dv = dq + 2 * ch;
  const float s2 = 1.0f / sqrtf((float)ch);   // (ch^-1/4)^2: the scale sits on both q and k
  auto stage = [&](int c0) {
    for (int e = threadIdx.x; e < T * 32; e += 256) {
      const int r = e >> 5, c = e & 31;
      const bool ok = c0 + c < ch;
      sq[r * 33 + c] = ok ? q[r * ld + c0 + c] : 0.f;
      sk[r * 33 + c] = ok ? k[r * ld + c0 + c] : 0.f;
      sv[r * 33 + c] = ok ? v[r * ld + c0 + c] : 0.f;
      sg[r * 33 + c] = ok ? go[r * ldo + c0 + c] : 0.f;
    }
  };
  // S = s2 q k^T and dP = dO v^T, accumulated over channel chunks (each thread owns entries e = tid + 256 n of the T x T matrices)
  for (int e = threadIdx.x; e < T * T; e += 256) { P[e] = 0.f; dS[e] = 0.f; }
  for (int c0 = 0; c0 < ch; c0 += 32) {
    __syncthreads();
    stage(c0);
    __syncthreads();
    for (int e = threadIdx.x; e < T * T; e += 256) {
      const int i = e / T, j = e - i * T;
      float acc = 0.f, accp = 0.f;
#pragma unroll 8
      for (int c = 0; c < 32; ++c) { acc += sq[i * 33 + c] * sk[j * 33 + c]; accp += sg[i * 33 + c] * sv[j * 33 + c]; }
      P[e] += acc; dS[e] += accp;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T; i += 256) {   // softmax row i, then dS = P * (dP - sum_j P*dP)
    float m = -INFINITY;
    for (int j = 0; j < T; ++j) { const float x = P[i * T + j] * s2; P[i * T + j] = x; m = fmaxf(m, x); }
    float s = 0.f;
    for (int j = 0; j < T; ++j) { const float e = __expf(P[i * T + j] - m); P[i * T + j] = e; s += e; }
    const float inv = 1.0f / s;
    float delta = 0.f;
    for (int j = 0; j < T; ++j) { const float p = P[i * T + j] * inv; P[i * T + j] = p; delta += p * dS[i * T + j]; }
    for (int j = 0; j < T; ++j) dS[i * T + j] = P[i * T + j] * (dS[i * T + j] - delta);
  }
  // dV[r] = sum_i P[i][r] dO[i]; dQ[r] = s2 sum_j dS[r][j] K[j]; dK[r] = s2 sum_i dS[i][r] Q[i], 32 channels at a time
  for (int c0 = 0; c0 < ch; c0 += 32) {
    __syncthreads();
    stage(c0);
    __syncthreads();
    for (int e = threadIdx.x; e < T * 32; e += 256) {
      const int r = e >> 5, c = e & 31;
      if (c0 + c >= ch) continue;
      float av = 0.f, aq = 0.f, ak = 0.f;
      for (int j = 0; j < T; ++j) {
        av += P[j * T + r] * sg[j * 33 + c];
        aq += dS[r * T + j] * sk[j * 33 + c];
        ak += dS[j * T + r] * sq[j * 33 + c];
      }
      dv[r * ld + c0 + c] = av; dq[r * ld + c0 + c] = aq * s2; dk[r * ld + c0 + c] = ak * s2;
    }
  }
}

// T = 64 tokens (the 8 x 8 level of the 32 x 32 latents): the same five contractions on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: both
// operands stay fp32, so every numerics mode shares it). One workgroup of 4 waves per (sample, head); q, k, v, dO are whole in LDS
// ([64][ch + 4] floats: the 16-B fragment reads of 16 consecutive rows cover the 64 banks once), P takes V's place after the first
// phase, dS its own [64][68] image.
//   phase 1: wave (ti, tj) owns the 32 x 32 tile of S = q k^T and dP = dO v^T; a lane (row r, half h) reads channels c0 + 4h .. + 3 of its A
//            and B rows as one float4 and feeds four MFMAs (MFMA n contracts channels c0 + n and c0 + 4 + n);
//   phase 2: softmax / dS rows by four lanes per row (xor-shuffles);
//   phase 3: dV = P^T dO, dQ = s2 dS K, dK = s2 dS^T Q: a wave takes (output, 32-channel tile) items, both 32-row tiles of an item share
//            the B reads.
typedef float att_f32x16 __attribute__((ext_vector_type(16)));
constexpr int A64_PP = 68;
__global__ void __launch_bounds__(256) attn64_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dO, float* __restrict__ dqkv,
                                                              int heads, int ch) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int PT = ch + 4;
  const int vrows = PT > A64_PP ? PT : A64_PP;
  float* sq = sm;
  float* sk = sq + 64 * PT;
  float* sg = sk + 64 * PT;
  float* sv = sg + 64 * PT;          // [64][PT], later P [64][68]
  float* sP = sv;
  float* sD = sv + 64 * vrows;       // dP, then dS [64][68]
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r = l & 31, hh = l >> 5;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const long ld = (long)heads * 3 * ch, ldo = (long)heads * ch;
  const float* q = qkv + (long)b * 64 * ld + (long)h * 3 * ch;
  const float* k = q + ch;
  const float* v = q + 2 * ch;
  const float* go = dO + (long)b * 64 * ldo + (long)h * ch;
  float* dq = dqkv + (long)b * 64 * ld + (long)h * 3 * ch;
  const float s2 = 1.0f / sqrtf((float)ch);
  const int c4n = ch >> 2;
#pragma unroll 2
  for (int idx = tid; idx < 64 * c4n; idx += 256) {
    const int row = idx / c4n, c = (idx - row * c4n) * 4;
    const float4 a = *reinterpret_cast<const float4*>(q + row * ld + c);
    const float4 bb = *reinterpret_cast<const float4*>(k + row * ld + c);
    const float4 cc = *reinterpret_cast<const float4*>(v + row * ld + c);
    const float4 dd = *reinterpret_cast<const float4*>(go + row * ldo + c);
    *reinterpret_cast<float4*>(sq + row * PT + c) = a;
    *reinterpret_cast<float4*>(sk + row * PT + c) = bb;
    *reinterpret_cast<float4*>(sv + row * PT + c) = cc;
    *reinterpret_cast<float4*>(sg + row * PT + c) = dd;
  }
  __syncthreads();
  {
    const int ti = w >> 1, tj = w & 1;
    att_f32x16 S, D;
#pragma unroll
    for (int e = 0; e < 16; ++e) { S[e] = 0.f; D[e] = 0.f; }
    const float* pa = sq + (32 * ti + r) * PT + 4 * hh;
    const float* pb = sk + (32 * tj + r) * PT + 4 * hh;
    const float* pc = sg + (32 * ti + r) * PT + 4 * hh;
    const float* pd = sv + (32 * tj + r) * PT + 4 * hh;
#pragma unroll 2
    for (int c0 = 0; c0 < ch; c0 += 8) {
      const float4 a = *reinterpret_cast<const float4*>(pa + c0), bb = *reinterpret_cast<const float4*>(pb + c0);
      const float4 g = *reinterpret_cast<const float4*>(pc + c0), vv = *reinterpret_cast<const float4*>(pd + c0);
      S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bb.x, S, 0, 0, 0);
      D = __builtin_amdgcn_mfma_f32_32x32x2f32(g.x, vv.x, D, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bb.y, S, 0, 0, 0);
      D = __builtin_amdgcn_mfma_f32_32x32x2f32(g.y, vv.y, D, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bb.z, S, 0, 0, 0);
      D = __builtin_amdgcn_mfma_f32_32x32x2f32(g.z, vv.z, D, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bb.w, S, 0, 0, 0);
      D = __builtin_amdgcn_mfma_f32_32x32x2f32(g.w, vv.w, D, 0, 0, 0);
    }
    __syncthreads();                 // every wave is done with V before P lands on it
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = 32 * ti + (e & 3) + 8 * (e >> 2) + 4 * hh;
      sP[row * A64_PP + 32 * tj + r] = S[e] * s2;
      sD[row * A64_PP + 32 * tj + r] = D[e];
    }
  }
  __syncthreads();
  {
    const int i = tid >> 2, qd = tid & 3;
    float* prow = sP + i * A64_PP + 16 * qd;
    float* drow = sD + i * A64_PP + 16 * qd;
    float x[16], dp[16];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const float4 t4 = *reinterpret_cast<const float4*>(prow + 4 * n), u4 = *reinterpret_cast<const float4*>(drow + 4 * n);
      x[4 * n] = t4.x; x[4 * n + 1] = t4.y; x[4 * n + 2] = t4.z; x[4 * n + 3] = t4.w;
      dp[4 * n] = u4.x; dp[4 * n + 1] = u4.y; dp[4 * n + 2] = u4.z; dp[4 * n + 3] = u4.w;
    }
    float m = x[0];
#pragma unroll
    for (int n = 1; n < 16; ++n) m = fmaxf(m, x[n]);
    m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2));
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < 16; ++n) { x[n] = __expf(x[n] - m); s += x[n]; }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
    const float inv = 1.0f / s;
    float delta = 0.f;
#pragma unroll
    for (int n = 0; n < 16; ++n) { x[n] *= inv; delta += x[n] * dp[n]; }
    delta += __shfl_xor(delta, 1); delta += __shfl_xor(delta, 2);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      *reinterpret_cast<float4*>(prow + 4 * n) = make_float4(x[4 * n], x[4 * n + 1], x[4 * n + 2], x[4 * n + 3]);
      *reinterpret_cast<float4*>(drow + 4 * n) = make_float4(x[4 * n] * (dp[4 * n] - delta), x[4 * n + 1] * (dp[4 * n + 1] - delta),
                                                             x[4 * n + 2] * (dp[4 * n + 2] - delta), x[4 * n + 3] * (dp[4 * n + 3] - delta));
    }
  }
  __syncthreads();
  const int nct = ch >> 5;
  for (int it = w; it < 3 * nct; it += 4) {
    const int ct = it % nct, o = it / nct;           // o: 0 dV (column offset 2 ch), 1 dQ (0), 2 dK (ch)
    // A element (m, kk): o = 0: P[kk][m]; 1: dS[m][kk]; 2: dS[kk][m]   ->   base[kk * ak + m * am]
    const float* abase = o == 0 ? sP : sD;
    const int ak = o == 1 ? 1 : A64_PP, am = o == 1 ? A64_PP : 1;
    const float* bbase = (o == 0 ? sg : (o == 1 ? sk : sq)) + 32 * ct + r;
    const float* a0 = abase + hh * ak + r * am;
    const float* a1 = a0 + 32 * am;
    const float* bp = bbase + hh * PT;
    att_f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll 8
    for (int k0 = 0; k0 < 64; k0 += 2) {
      const float bv = bp[k0 * PT];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[k0 * ak], bv, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[k0 * ak], bv, acc1, 0, 0, 0);
    }
    const float sc = o == 0 ? 1.0f : s2;
    float* dst = dq + (o == 0 ? 2 * ch : (o == 1 ? 0 : ch)) + 32 * ct + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * hh;
      dst[(long)row * ld] = acc0[e] * sc;
      dst[(long)(row + 32) * ld] = acc1[e] * sc;
    }
  }
}

// General-T form (T too large for both T x T matrices in LDS): two kernels over a global scratch [B*heads][2][T][T] (P, then dS).
// A: grid (B*heads, ceil(T/QT)) — the QT query rows of this block against all keys: S, dP in LDS, softmax, dS, dQ; P and dS rows go to
//    the scratch. B: grid (B*heads, ceil(T/32)) — 32 key rows: dV = P^T dO, dK = s2 dS^T Q reading scratch columns.
__global__ void __launch_bounds__(256) attn_bwd_rows_kernel(const float* __restrict__ qkv, const float* __restrict__ dO, float* __restrict__ dqkv,
                                                            float* __restrict__ scratch, int T, int heads, int ch, int QT) {
  extern __shared__ float sm[];
  float* P = sm;                    // [QT][T]
  float* dS = sm + (long)QT * T;    // [QT][T]
  const int bh = blockIdx.x, b = bh / heads, h = bh % heads, i0 = blockIdx.y * QT;
  const int nq = min(QT, T - i0);
  const long ld = (long)heads * 3 * ch, ldo = (long)heads * ch;
  const float* q = qkv + (long)b * T * ld + (long)h * 3 * ch;
  const float* k = q + ch;
  const float* v = q + 2 * ch;
  const float* go = dO + (long)b * T * ldo + (long)h * ch;
  float* dq = dqkv + (long)b * T * ld + (long)h * 3 * ch;
  const float s2 = 1.0f / sqrtf((float)ch);
  for (int e = threadIdx.x; e < nq * T; e += 256) {
    const int il = e / T, j = e - il * T, i = i0 + il;
    float acc = 0.f, accp = 0.f;
    for (int c = 0; c < ch; c += 4) {
      const float4 a4 = *reinterpret_cast<const float4*>(q + i * ld + c), b4 = *reinterpret_cast<const float4*>(k + j * ld + c);
      acc += a4.x * b4.x + a4.y * b4.y + a4.z * b4.z + a4.w * b4.w;
      const float4 g4 = *reinterpret_cast<const float4*>(go + i * ldo + c), v4 = *reinterpret_cast<const float4*>(v + j * ld + c);
      accp += g4.x * v4.x + g4.y * v4.y + g4.z * v4.z + g4.w * v4.w;
    }
    P[il * T + j] = acc * s2; dS[il * T + j] = accp;
  }
  __syncthreads();
  for (int il = threadIdx.x; il < nq; il += 256) {
    float m = -INFINITY;
    for (int j = 0; j < T; ++j) m = fmaxf(m, P[il * T + j]);
    float s = 0.f;
    for (int j = 0; j < T; ++j) { const float e = __expf(P[il * T + j] - m); P[il * T + j] = e; s += e; }
    const float inv = 1.0f / s;
    float delta = 0.f;
    for (int j = 0; j < T; ++j) { const float pp = P[il * T + j] * inv; P[il * T + j] = pp; delta += pp * dS[il * T + j]; }
    for (int j = 0; j < T; ++j) dS[il * T + j] = P[il * T + j] * (dS[il * T + j] - delta);
  }
  __syncthreads();
  float* sp = scratch + (long)bh * 2 * T * T;
  for (int e = threadIdx.x; e < nq * T; e += 256) {
    const int il = e / T, j = e - il * T;
    sp[(long)(i0 + il) * T + j] = P[e];
    sp[(long)T * T + (long)(i0 + il) * T + j] = dS[e];
  }
  for (int e = threadIdx.x; e < nq * ch; e += 256) {
    const int il = e / ch, c = e - il * ch;
    float aq = 0.f;
    for (int j = 0; j < T; ++j) aq += dS[il * T + j] * k[j * ld + c];
    dq[(long)(i0 + il) * ld + c] = aq * s2;
  }
}

__global__ void __launch_bounds__(256) attn_bwd_cols_kernel(const float* __restrict__ qkv, const float* __restrict__ dO, float* __restrict__ dqkv,
                                                            const float* __restrict__ scratch, int T, int heads, int ch) {
  const int bh = blockIdx.x, b = bh / heads, h = bh % heads, j0 = blockIdx.y * 32;
  const int nk = min(32, T - j0);
  const long ld = (long)heads * 3 * ch, ldo = (long)heads * ch;
  const float* q = qkv + (long)b * T * ld + (long)h * 3 * ch;
  const float* go = dO + (long)b * T * ldo + (long)h * ch;
  float* dk = dqkv + (long)b * T * ld + (long)h * 3 * ch + ch;
  float* dv = dk + ch;
  const float* Pm = scratch + (long)bh * 2 * T * T;
  const float* dSm = Pm + (long)T * T;
  const float s2 = 1.0f / sqrtf((float)ch);
  for (int e = threadIdx.x; e < nk * ch; e += 256) {
    const int jl = e / ch, c = e - jl * ch, j = j0 + jl;
    float av = 0.f, ak = 0.f;
    for (int i = 0; i < T; ++i) {
      av += Pm[(long)i * T + j] * go[i * ldo + c];
      ak += dSm[(long)i * T + j] * q[i * ld + c];
    }
    dv[(long)j * ld + c] = av; dk[(long)j * ld + c] = ak * s2;
  }
}

// ------------------------------------------------------------------------------------------------ small fp32 GEMM
// C[M][N] = alpha * op(A) op(B) + beta * C; op(A)[m][k] = ta ? A[k*lda + m] : A[m*lda + k]; op(B)[k][n] = tb ? B[n*ldb + k] : B[k*ldb + n]
// (the tiled kernel is sgemm.hpp's; this unit holds the fixed-order reduce of its split-K slices)
__global__ void gemm_f32_reduce_kernel(const float* __restrict__ part, int ks, float* __restrict__ Cm, long ldc, int M, int N, float alpha, float beta) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)M * N) return;
  float s = 0.f;
  for (int z = 0; z < ks; ++z) s += part[(long)z * M * N + i];   // fixed order
  const long o = (i / N) * ldc + (i % N);
  Cm[o] = alpha * s + (beta != 0.f ? beta * Cm[o] : 0.f);
}

// mode 0: out = silu(x); mode 1: out = dy * silu'(x)
__global__ void silu_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, long n, int mode) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = mode == 0 ? silu_f(x[i]) : dy[i] * silu_grad(x[i]);
}

// y = alpha * x + beta * y (gradient accumulation over micro-batches: accumulate_grad_batches of the reference's Trainer)
__global__ void axpby_kernel(const float* __restrict__ x, float* __restrict__ y, long n4, float alpha, float beta) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 a = reinterpret_cast<const float4*>(x)[i];
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (beta != 0.f) b = reinterpret_cast<float4*>(y)[i];      // beta = 0 starts an accumulation window: y may hold anything (NaN included), as in BLAS
  b.x = alpha * a.x + beta * b.x; b.y = alpha * a.y + beta * b.y; b.z = alpha * a.z + beta * b.z; b.w = alpha * a.w + beta * b.w;
  reinterpret_cast<float4*>(y)[i] = b;
}

// q_sample (ddpm.py:277-280 + extract_into_tensor util.py:96-99): out = sqrt_ac[t[b]] * x0 + sqrt_1mac[t[b]] * noise; n = elements per sample
__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const int64_t* __restrict__ t, const float* __restrict__ sa,
                                const float* __restrict__ s1, float* __restrict__ out, long n, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t tb = t[i / n];
  out[i] = __fadd_rn(__fmul_rn(sa[tb], x0[i]), __fmul_rn(s1[tb], noise[i]));     // two products then a sum, like the reference (no FMA contraction)
}

// ------------------------------------------------------------------------------------------------ loss
// L1 (ddpm.py:282-295, 1030-1040): loss = mean |target - pred|; dpred = sign(pred - target) * scale / n
__global__ void __launch_bounds__(256) l1_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target, float* __restrict__ dpred, long n,
                                                         float gscale, double* __restrict__ part) {
  __shared__ double red[256];
  double s = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float d = pred[i] - target[i];
    s += (double)fabsf(d);
    if (dpred) dpred[i] = d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void l1_final_kernel(const double* __restrict__ part, int nb, double inv_n, float* __restrict__ loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nb; ++i) s += part[i];
    *loss = (float)(s * inv_n);
  }
}

// ------------------------------------------------------------------------------------------------ optimizer
// AdamW (torch.optim.AdamW semantics) + LitEma (ema.py:25-44) over many tensors in one launch. table[t] = {p, g, m, v, ema, n};
// chunk_tensor[blockIdx.x] / chunk_off[blockIdx.x] map a block to 4096 elements of one tensor.
struct OptTensor { float* p; const float* g; float* m; float* v; float* ema; long n; };
// one element of torch.optim.AdamW (decoupled weight decay first, bias corrections bc1 = 1 - beta1^t, bc2_sqrt = sqrt(1 - beta2^t))
__device__ __forceinline__ void adamw_one(float& p, float g_raw, float& m, float& v, float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                          float bc2_sqrt, float grad_scale) {
  const float g = g_raw * grad_scale;
  float pn = p * (1.0f - lr * wd);
  m = beta1 * m + (1.0f - beta1) * g;
  v = beta2 * v + (1.0f - beta2) * g * g;
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  pn -= (lr / bc1) * (m / denom);
  p = pn;
}
__global__ void __launch_bounds__(256) adamw_ema_kernel(const OptTensor* __restrict__ table, const int* __restrict__ chunk_tensor,
                                                        const long* __restrict__ chunk_off, float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                        float bc2_sqrt, float ema_decay, float grad_scale, const float* __restrict__ sched,
                                                        const int* __restrict__ sched_idx) {
  if (sched) {      // graph replay: this step's row of the host-built schedule {bc1, sqrt(bc2), ema_decay, lr} (stedm_adamw_ema_sched)
    const float* r = sched + 4 * (long)*sched_idx;
    bc1 = r[0]; bc2_sqrt = r[1]; ema_decay = r[2]; lr = r[3];
  }
  const OptTensor t = table[chunk_tensor[blockIdx.x]];
  const long o0 = chunk_off[blockIdx.x];
  const uintptr_t al = reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) | reinterpret_cast<uintptr_t>(t.m) |
                       reinterpret_cast<uintptr_t>(t.v) | reinterpret_cast<uintptr_t>(t.ema);
  if (o0 + 4096 <= t.n && (al & 15) == 0) {      // a whole chunk of 16-B aligned streams: 16 B per lane and access
    const bool has_e = t.ema != nullptr;
#pragma unroll 2
    for (int k = 0; k < 4; ++k) {
      const long i = o0 + (k * 256 + threadIdx.x) * 4;
      const float4 p4 = *reinterpret_cast<const float4*>(t.p + i), g4 = *reinterpret_cast<const float4*>(t.g + i);
      const float4 m4 = *reinterpret_cast<const float4*>(t.m + i), v4 = *reinterpret_cast<const float4*>(t.v + i);
      float4 e4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_e) e4 = *reinterpret_cast<const float4*>(t.ema + i);
      float pv[4] = {p4.x, p4.y, p4.z, p4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
      const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) adamw_one(pv[j], gv[j], mv[j], vv[j], lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, grad_scale);
      *reinterpret_cast<float4*>(t.p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      *reinterpret_cast<float4*>(t.m + i) = make_float4(mv[0], mv[1], mv[2], mv[3]);
      *reinterpret_cast<float4*>(t.v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      if (has_e) {
        float ev[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) ev[j] = ev[j] - (1.0f - ema_decay) * (ev[j] - pv[j]);
        *reinterpret_cast<float4*>(t.ema + i) = make_float4(ev[0], ev[1], ev[2], ev[3]);
      }
    }
    return;
  }
  for (int k = 0; k < 16; ++k) {
    const long i = o0 + k * 256 + threadIdx.x;
    if (i >= t.n) return;
    float p = t.p[i], m = t.m[i], v = t.v[i];
    adamw_one(p, t.g[i], m, v, lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, grad_scale);
    t.p[i] = p; t.m[i] = m; t.v[i] = v;
    if (t.ema) { const float s = t.ema[i]; t.ema[i] = s - (1.0f - ema_decay) * (s - p); }
  }
}

// AdamW + EMA for convolution weights that ALSO writes the 16-bit fragment-order copies the next forward / backward read (the packs of
// stedm_pack_frag_multi: forward order and the flipped / transposed dgrad order, 32x32x16 and 16x16x32 fragment forms), so the optimizer's own
// pass over the weights replaces the re-pack launches and their extra read. A block owns a 32 (cout) x 32 (cin) x taps piece of an OIHW
// filter: rows of 32 * taps contiguous floats in, the updated values kept in LDS, and every fragment vector (8 consecutive channels of one
// row and tap) that lies inside the piece written from there — a piece holds whole fragments of all four forms.
struct FusedPackOut { void* out; int transposed, flip, m16, f16; };
struct FusedOptDesc {
  float* p; const float* g; float* m; float* v; float* ema;
  int cout, cin, taps, blk0, nout, pad_;
  FusedPackOut o[4];
};
static_assert(sizeof(FusedPackOut) == 24 && sizeof(FusedOptDesc) == 160, "FusedOptDesc layout is part of the ABI (stedm_adamw_ema_pack)");

template <int ROWS, int CIW>
__global__ void __launch_bounds__(256) adamw_ema_pack_kernel(const FusedOptDesc* __restrict__ descs, const int nd, float lr, float beta1, float beta2,
                                                             float eps, float wd, float bc1, float bc2_sqrt, float ema_decay, float grad_scale,
                                                             const float* __restrict__ sched, const int* __restrict__ sched_idx) {
  extern __shared__ float ftile[];        // [ROWS][CIW * taps + 1]
  if (sched) {
    const float* r = sched + 4 * (long)*sched_idx;
    bc1 = r[0]; bc2_sqrt = r[1]; ema_decay = r[2]; lr = r[3];
  }
  int lo = 0, hi = nd - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const FusedOptDesc& d = descs[lo];
  const int bid = blockIdx.x - d.blk0;
  const int taps = d.taps, cin = d.cin, cout = d.cout;
  const int ncb = cin / CIW;
  const int co0 = (bid / ncb) * ROWS, ci0 = (bid % ncb) * CIW;
  const int run = CIW * taps, pitch = run + 1, r4 = run >> 2;
  float* const P = d.p; const float* const G = d.g; float* const M = d.m; float* const V = d.v; float* const E = d.ema;
  const bool has_e = E != nullptr;
#pragma unroll 2
  for (int idx = threadIdx.x; idx < ROWS * r4; idx += 256) {
    const int row = idx / r4, r0 = (idx - row * r4) * 4;
    const long off = ((long)(co0 + row) * cin + ci0) * taps + r0;
    const float4 p4 = *reinterpret_cast<const float4*>(P + off), g4 = *reinterpret_cast<const float4*>(G + off);
    const float4 m4 = *reinterpret_cast<const float4*>(M + off), v4 = *reinterpret_cast<const float4*>(V + off);
    float4 e4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_e) e4 = *reinterpret_cast<const float4*>(E + off);
    float pv[4] = {p4.x, p4.y, p4.z, p4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
    const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) adamw_one(pv[j], gv[j], mv[j], vv[j], lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, grad_scale);
    *reinterpret_cast<float4*>(P + off) = make_float4(pv[0], pv[1], pv[2], pv[3]);
    *reinterpret_cast<float4*>(M + off) = make_float4(mv[0], mv[1], mv[2], mv[3]);
    *reinterpret_cast<float4*>(V + off) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    if (has_e) {
      float ev[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) ev[j] = ev[j] - (1.0f - ema_decay) * (ev[j] - pv[j]);
      *reinterpret_cast<float4*>(E + off) = make_float4(ev[0], ev[1], ev[2], ev[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) ftile[row * pitch + r0 + j] = pv[j];
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int oi = 0; oi < d.nout; ++oi) {
    const FusedPackOut po = d.o[oi];
    const int fr = po.m16 ? 16 : 32, fc = po.m16 ? 32 : 16;                 // fragment rows x channels
    const int Rn = po.transposed ? CIW : ROWS, Rc = po.transposed ? ROWS : CIW;   // the piece in the pack's (row n, channel c) coordinates
    const int n_base = po.transposed ? ci0 : co0, c_base = po.transposed ? co0 : ci0;
    const int nch = (po.transposed ? cout : cin) / fc;                       // channel chunks of the pack
    const int nfn = Rn / fr, nfrag = nfn * (Rc / fc) * taps;
    const int g = lane >> 4;
    for (int f = wave; f < nfrag; f += 4) {
      const int tap = f % taps, ff = f / taps, fn = ff % nfn, fcx = ff / nfn;
      int nl, cl;
      if (!po.m16) { nl = fn * 32 + (lane & 31); cl = fcx * 16 + (lane >> 5) * 8; }
      else { nl = fn * 16 + (lane & 15); cl = fcx * 32 + 8 * (taps == 9 ? 2 * (g & 1) + (g >> 1) : g); }
      const int ts = po.flip ? taps - 1 - tap : tap;        // source tap
      const float* src = po.transposed ? ftile + cl * pitch + nl * taps + ts : ftile + nl * pitch + cl * taps + ts;
      const int es = po.transposed ? pitch : taps;
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = src[e * es];
      const int n = n_base + nl, c = c_base + cl;
      const long at = po.m16 ? ((((long)(n >> 7) * nch + (c >> 5)) * taps + tap) * 8 + ((n >> 4) & 7)) * 512 + ((n & 15) + 16 * g) * 8
                             : ((((long)(n >> 7) * nch + (c >> 4)) * taps + tap) * 4 + ((n >> 5) & 3)) * 512 + ((n & 31) + 32 * ((c >> 3) & 1)) * 8;
      if (po.f16) {
        typedef _Float16 H8 __attribute__((ext_vector_type(8)));
        H8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)x[e];
        *reinterpret_cast<H8*>(reinterpret_cast<_Float16*>(po.out) + at) = o;
      } else {
        typedef __bf16 B8 __attribute__((ext_vector_type(8)));
        B8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)x[e];
        *reinterpret_cast<B8*>(reinterpret_cast<__bf16*>(po.out) + at) = o;
      }
    }
  }
}


// LitEma.forward alone (ema.py:25-44, called from on_train_batch_end, ddpm.py:369-371): shadow -= (1 - decay) * (shadow - p) over the
// optimizer's pointer table; tensors without a shadow are skipped.
__global__ void __launch_bounds__(256) ema_update_kernel(const OptTensor* __restrict__ table, const int* __restrict__ chunk_tensor,
                                                         const long* __restrict__ chunk_off, float one_minus_decay) {
  const OptTensor t = table[chunk_tensor[blockIdx.x]];
  if (!t.ema) return;
  const long o0 = chunk_off[blockIdx.x];
  for (int k = 0; k < 16; ++k) {
    const long i = o0 + k * 256 + threadIdx.x;
    if (i >= t.n) return;
    const float s = t.ema[i];
    t.ema[i] = s - one_minus_decay * (s - t.p[i]);
  }
}

// SpatialRescaler (encoders/modules.py:123-130) weight gradient: dW[co][ci] = sum_{b,p} d_out[b][co][p] * boxmean_f(x)[b][ci][p].
// grid B (one block per sample, per-sample partials), then a fixed-order sum over the batch.
__global__ void __launch_bounds__(256) rescale_wgrad_partial_kernel(const float* __restrict__ x, const float* __restrict__ d_out, float* __restrict__ part, int cin,
                                                                    int cout, int H, int W, int f) {
  __shared__ float red[256];
  const int b = blockIdx.x, Ho = H / f, Wo = W / f, np = Ho * Wo;
  for (int e = 0; e < cin * cout; ++e) {
    const int co = e / cin, ci = e % cin;
    float s = 0.f;
    for (int p = threadIdx.x; p < np; p += 256) {
      const int yo = p / Wo, xo = p % Wo;
      const float* px = x + (((long)b * cin + ci) * H + (long)yo * f) * W + (long)xo * f;
      float m = 0.f;
      for (int dy = 0; dy < f; ++dy)
        for (int dx = 0; dx < f; ++dx) m += px[(long)dy * W + dx];
      s += d_out[((long)b * cout + co) * np + p] * (m / (float)(f * f));
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) part[(long)b * cin * cout + e] = red[0];
    __syncthreads();
  }
}
__global__ void rescale_wgrad_final_kernel(const float* __restrict__ part, int B, int n, float* __restrict__ dw, int accumulate) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += part[(long)b * n + e];
  dw[e] = (accumulate ? dw[e] : 0.f) + s;
}

}  // namespace

// ================================================================================================ C ABI
extern "C" int stedm_spatial_rescale_wgrad(const float* x, const float* d_out, float* ws, float* dw, int B, int cin, int cout, int H, int W, int n_stages,
                                           int accumulate, void* stream) {
  STEDM_CHECK_ARG(x && d_out && ws && dw && n_stages >= 0 && n_stages < 8 && cin * cout <= 1024, "spatial_rescale_wgrad: bad args");
  const int f = 1 << n_stages;
  STEDM_CHECK_ARG(H % f == 0 && W % f == 0, "spatial_rescale_wgrad: H, W must be divisible by 2^n_stages");
  rescale_wgrad_partial_kernel<<<B, 256, 0, as_stream(stream)>>>(x, d_out, ws, cin, cout, H, W, f);
  rescale_wgrad_final_kernel<<<(cin * cout + 63) / 64, 64, 0, as_stream(stream)>>>(ws, B, cin * cout, dw, accumulate);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_gn_fold(const float* cs1, int nslab1, int c1, const float* cs2, int nslab2, int c2, int groups, int B, int HW, float eps, float* mean_rstd,
                             void* stream) {
  STEDM_CHECK_ARG(cs1 && mean_rstd && groups > 0 && groups <= 64 && (c1 + c2) % groups == 0 && (cs2 != nullptr) == (c2 > 0), "gn_fold: bad args");
  gn_fold_kernel<<<B, 256, 0, as_stream(stream)>>>(cs1, nslab1, c1, cs2, nslab2, c2, groups, HW, eps, mean_rstd);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_gn_bwd(const float* x1, int c1, const float* x2, int c2, const float* mean_rstd, const float* gamma, const float* beta, int groups, int act,
                            const float* dA, const float* add, int B, int HW, float* ws, float* dx1, int acc1, float* dx2, int acc2, void* dx16_hi,
                            void* dx16_lo, int mm_dtype, float* dgamma, float* dbeta, int acc_param, void* stream) {
  const int C = c1 + c2;
  STEDM_CHECK_ARG(x1 && mean_rstd && gamma && beta && dA && ws && dx1 && dgamma && dbeta, "gn_bwd: null pointer");
  STEDM_CHECK_ARG((x2 != nullptr) == (c2 > 0) && (c2 == 0 || dx2), "gn_bwd: x2/c2/dx2 mismatch");
  STEDM_CHECK_ARG(C % 4 == 0 && c1 % 4 == 0 && groups > 0 && groups <= 64 && C % groups == 0 && C * 8 <= 65536, "gn_bwd: channel constraints");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "gn_bwd: bad mm_dtype");
  const int nslab = (HW + kGnBwdSlab - 1) / kGnBwdSlab, Q = C / 4;
  // workspace: part [B][nslab][C][2] | bc [B][C][2] | gm [B][groups][2]
  float* part = ws;
  float* bc = part + (long)B * nslab * C * 2;
  float* gm = bc + (long)B * C * 2;
  GnBwdArgs a{x1, x2, c1, c2, mean_rstd, gamma, beta, dA, groups, HW, act, part, gm, add, dx1, dx2, acc1, acc2, dx16_hi, dx16_lo};
  hipStream_t st = as_stream(stream);
  int ppt = 0;
  const int cb = gn_bwd_fused_cb(c1, c2, groups, HW, &ppt);
  if (cb > 0) {     // one pass: x and dA read once, no slab partials
    GnBwdFusedArgs fa{a, bc, cb};
    if (mm_dtype == STEDM_F16) launch_gn_bwd_fused<_Float16>(fa, ppt, dim3(B, C / cb), st);
    else launch_gn_bwd_fused<__bf16>(fa, ppt, dim3(B, C / cb), st);
    gn_bwd_param_kernel<<<(C + 15) / 16, 256, 0, st>>>(bc, B, C, dgamma, dbeta, acc_param);
    STEDM_LAUNCH_CHECK();
    return 0;
  }
  gn_bwd_stats_kernel<<<dim3(B, nslab, (Q + 63) / 64), 256, 0, st>>>(a);
  gn_bwd_fold_kernel<<<B, 256, (size_t)C * 8, st>>>(part, nslab, C, groups, HW, gamma, bc, gm);
  gn_bwd_param_kernel<<<(C + 15) / 16, 256, 0, st>>>(bc, B, C, dgamma, dbeta, acc_param);
  const dim3 grid(B, (unsigned)(((long)HW * Q + 1023) / 1024));
  if (mm_dtype == STEDM_F16) gn_bwd_apply_kernel<_Float16><<<grid, 256, 0, st>>>(a);
  else gn_bwd_apply_kernel<__bf16><<<grid, 256, 0, st>>>(a);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" long stedm_gn_bwd_ws_floats(int B, int HW, int C, int groups) {
  return (long)B * ((HW + kGnBwdSlab - 1) / kGnBwdSlab) * C * 2 + (long)B * C * 2 + (long)B * groups * 2;
}

extern "C" int stedm_im2col_t16(const void* src16, void* dst16, int B, int Hs, int Ws, int C, int ks, int mode, long Ppad, void* stream) {
  STEDM_CHECK_ARG(src16 && dst16 && C % 8 == 0 && (ks == 1 || ks == 3) && mode >= 0 && mode <= 2 && Ppad % 64 == 0, "im2col_t16: bad args");
  const int Ho = mode == 1 ? 2 * Hs : (mode == 2 ? Hs / 2 : Hs), Wo = mode == 1 ? 2 * Ws : (mode == 2 ? Ws / 2 : Ws);
  const long P = (long)B * Ho * Wo;
  STEDM_CHECK_ARG(Ppad >= P, "im2col_t16: Ppad < P");
  dim3 grid((unsigned)(Ppad / 64), (C + 63) / 64, ks * ks);
  im2col_t16_kernel<<<grid, 256, 0, as_stream(stream)>>>((const uint16_t*)src16, (uint16_t*)dst16, B, Hs, Ws, C, Ho, Wo, ks, mode, P, Ppad);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_wgrad_to_oihw(const float* dw, float* grad, int cout, int cin, int taps, int cin_ld, int cout_ld, int accumulate, int nsplit,
                                   void* stream) {
  STEDM_CHECK_ARG(dw && grad && cin_ld >= cin && cout_ld >= cout && nsplit >= 1, "wgrad_to_oihw: bad args");
  STEDM_CHECK_ARG(taps >= 1 && taps <= 9, "wgrad_to_oihw: taps must be 1..9");
  if (((cout + 31) / 32) * ((cin + 15) / 16) < 128 && nsplit >= 4)     // few blocks, long serial walks: quarter tiles
    wgrad_to_oihw_kernel<4><<<dim3((cout + 31) / 32, (cin + 3) / 4), 256, 0, as_stream(stream)>>>(dw, grad, cout, cin, taps, cin_ld, cout_ld, accumulate, nsplit);
  else
    wgrad_to_oihw_kernel<16><<<dim3((cout + 31) / 32, (cin + 15) / 16), 256, 0, as_stream(stream)>>>(dw, grad, cout, cin, taps, cin_ld, cout_ld, accumulate, nsplit);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_chan_sum_fold2(const float* cs, int B, int nslab, int C, float* per_sample, long ld, float* total, int accumulate, float* total2,
                                    void* stream) {
  STEDM_CHECK_ARG(cs && (per_sample || total) && (total || !total2), "chan_sum_fold: bad args");
  chan_sum_fold_kernel<<<(C + 15) / 16, 256, 0, as_stream(stream)>>>(cs, B, nslab, C, per_sample, ld, total, accumulate, total2);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_chan_sum_fold(const float* cs, int B, int nslab, int C, float* per_sample, long ld, float* total, int accumulate, void* stream) {
  return stedm_chan_sum_fold2(cs, B, nslab, C, per_sample, ld, total, accumulate, nullptr, stream);
}

extern "C" int stedm_sum2x2(const float* in, float* out, int B, int H, int W, int C, int accumulate, void* stream) {
  STEDM_CHECK_ARG(in && out && C % 4 == 0, "sum2x2: bad args");
  const long total = (long)B * H * W * (C / 4);
  sum2x2_kernel<<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(in, out, H, W, C / 4, accumulate, total);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_zero_insert16(const float* in, void* hi, void* lo, int B, int Ho, int Wo, int C, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(in && hi && C % 4 == 0 && (mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16), "zero_insert16: bad args");
  const long total = (long)B * 4 * Ho * Wo * (C / 4);
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (mm_dtype == STEDM_F16) zero_insert16_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>(in, (_Float16*)hi, (_Float16*)lo, Ho, Wo, C / 4, total);
  else zero_insert16_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>(in, (__bf16*)hi, (__bf16*)lo, Ho, Wo, C / 4, total);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" long stedm_attn_legacy_bwd_ws_floats(int B, int T, int heads) {
  const size_t lds = (size_t)T * T * 8 + (size_t)4 * T * 33 * 4;
  return lds <= 160 * 1024 - 1024 ? 0 : (long)B * heads * 2 * T * T;     // the LDS-resident form needs no workspace
}

extern "C" int stedm_attn_legacy_bwd(const float* qkv, const float* d_out, float* d_qkv, int B, int T, int heads, int ch, float* ws, void* stream) {
  STEDM_CHECK_ARG(qkv && d_out && d_qkv && ch % 4 == 0, "attn_legacy_bwd: bad args");
  const size_t lds = (size_t)T * T * 8 + (size_t)4 * T * 33 * 4;
  static bool attr = false;
  if (!attr) {
    STEDM_HIP_TRY(hipFuncSetAttribute((const void*)attn_legacy_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    STEDM_HIP_TRY(hipFuncSetAttribute((const void*)attn_bwd_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    attr = true;
  }
  const size_t lds64 = ((size_t)3 * 64 * (ch + 4) + (size_t)64 * (ch + 4 > A64_PP ? ch + 4 : A64_PP) + (size_t)64 * A64_PP) * sizeof(float);
  if (T == 64 && ch % 32 == 0 && lds64 <= 160 * 1024 - 1024 &&
      ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0) {
    static bool attr64 = false;
    if (!attr64) {
      STEDM_HIP_TRY(hipFuncSetAttribute((const void*)attn64_bwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
      attr64 = true;
    }
    attn64_bwd_mfma_kernel<<<B * heads, 256, lds64, as_stream(stream)>>>(qkv, d_out, d_qkv, heads, ch);
  } else if (lds <= 160 * 1024 - 1024) {
    attn_legacy_bwd_kernel<<<B * heads, 256, lds, as_stream(stream)>>>(qkv, d_out, d_qkv, T, heads, ch);
  } else {
    STEDM_CHECK_ARG(ws, "attn_legacy_bwd: T = %d tokens need the workspace of stedm_attn_legacy_bwd_ws_floats", T);
    int QT = 32;
    while (QT > 1 && (size_t)QT * T * 8 > 128 * 1024) QT >>= 1;
    STEDM_CHECK_ARG((size_t)QT * T * 8 <= 159 * 1024, "attn_legacy_bwd: T = %d tokens too many", T);
    attn_bwd_rows_kernel<<<dim3(B * heads, (T + QT - 1) / QT), 256, (size_t)QT * T * 8, as_stream(stream)>>>(qkv, d_out, d_qkv, ws, T, heads, ch, QT);
    attn_bwd_cols_kernel<<<dim3(B * heads, (T + 31) / 32), 256, 0, as_stream(stream)>>>(qkv, d_out, d_qkv, ws, T, heads, ch);
  }
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_gemm_f32(const float* A, long lda, int trans_a, const float* Bm, long ldb, int trans_b, float* Cm, long ldc, int M, int N, int K, float alpha,
                              float beta, float* ws, long ws_floats, void* stream) {
  STEDM_CHECK_ARG(A && Bm && Cm && M > 0 && N > 0 && K > 0, "gemm_f32: bad args");
  const int tiles = ((N + 63) / 64) * ((M + 63) / 64);
  int ks = 1;
  if (ws && tiles < 128 && K >= 1024) {          // few output tiles, long K: slices of K on separate blocks, fixed-order reduce
    ks = K / 256 < 64 ? K / 256 : 64;
    while (ks > 1 && (long)ks * M * N > ws_floats) --ks;
  }
  const int kchunk = ks > 1 ? ((K + ks - 1) / ks + 31) / 32 * 32 : K;
  if (ks > 1) ks = (K + kchunk - 1) / kchunk;
  stedm::SgemmArgs g{A, lda, trans_a, Bm, ldb, trans_b, Cm, ldc, M, N, K, alpha, beta, kchunk, ws, nullptr, 0, 0};
  stedm::sgemm_launch(g, ks, as_stream(stream));
  if (ks > 1) gemm_f32_reduce_kernel<<<(unsigned)(((long)M * N + 255) / 256), 256, 0, as_stream(stream)>>>(ws, ks, Cm, ldc, M, N, alpha, beta);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_silu(const float* x, const float* dy, float* out, long n, int mode, void* stream) {
  STEDM_CHECK_ARG(x && out && (mode == 0 || dy), "silu: bad args");
  silu_kernel<<<(unsigned)((n + 255) / 256), 256, 0, as_stream(stream)>>>(x, dy, out, n, mode);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_axpby_f32(const float* x, float* y, long n, float alpha, float beta, void* stream) {
  STEDM_CHECK_ARG(x && y && n > 0 && n % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0), "axpby_f32: n %% 4 and 16-B alignment required");
  axpby_kernel<<<(unsigned)((n / 4 + 255) / 256), 256, 0, as_stream(stream)>>>(x, y, n / 4, alpha, beta);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac, const float* sqrt_1mac, float* out, int B, long n,
                              void* stream) {
  STEDM_CHECK_ARG(x0 && noise && t && sqrt_ac && sqrt_1mac && out && B > 0 && n > 0, "q_sample: bad args");
  const long total = (long)B * n;
  q_sample_kernel<<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(x0, noise, t, sqrt_ac, sqrt_1mac, out, n, total);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_l1_loss(const float* pred, const float* target, long n, float grad_scale, float* d_pred, double* ws, float* loss, void* stream) {
  STEDM_CHECK_ARG(pred && target && ws && loss && n > 0, "l1_loss: bad args");
  const int nb = (int)((n + 256 * 16 - 1) / (256 * 16) < 1024 ? (n + 256 * 16 - 1) / (256 * 16) : 1024);
  l1_partial_kernel<<<nb, 256, 0, as_stream(stream)>>>(pred, target, d_pred, n, grad_scale / (float)n, ws);
  l1_final_kernel<<<1, 64, 0, as_stream(stream)>>>(ws, nb, 1.0 / (double)n, loss);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_adamw_ema(const void* table, const int* chunk_tensor, const long* chunk_off, int nchunks, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int step, float ema_decay, float grad_scale, void* stream) {
  STEDM_CHECK_ARG(table && chunk_tensor && chunk_off && nchunks > 0 && step >= 1, "adamw_ema: bad args");
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  adamw_ema_kernel<<<nchunks, 256, 0, as_stream(stream)>>>((const OptTensor*)table, chunk_tensor, chunk_off, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2),
                                                          ema_decay, grad_scale, nullptr, nullptr);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// The two optimizer passes with the step-dependent scalars read on the device: sched [n][4] = {1 - beta1^s, sqrt(1 - beta2^s), ema decay of step s,
// learning rate of step s} built by the host for a window of steps, *sched_idx = the row of the step being replayed (advanced by stedm_step_advance
// inside the captured step). A captured training step (UNetTrainer.capture_step) replays with nothing but that index changing.
extern "C" int stedm_adamw_ema_sched(const void* table, const int* chunk_tensor, const long* chunk_off, int nchunks, float beta1, float beta2, float eps,
                                     float weight_decay, const float* sched, const int* sched_idx, float grad_scale, void* stream) {
  STEDM_CHECK_ARG(table && chunk_tensor && chunk_off && nchunks > 0 && sched && sched_idx, "adamw_ema_sched: bad args");
  adamw_ema_kernel<<<nchunks, 256, 0, as_stream(stream)>>>((const OptTensor*)table, chunk_tensor, chunk_off, 0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f, 0.f,
                                                          grad_scale, sched, sched_idx);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// piece geometry of stedm_adamw_ema_pack: a tensor takes (cout / rows) * (cin / ciw) blocks. 32 x 32, 64 x 32, 32 x 64 and 64 x 64 measured the
// same on the north-star U-Net: the pass is bound by its 40 bytes per weight (234.6 M weights in 1.72 ms = 5.4 TB/s; the plain kernel moves
// its 36 bytes per weight at 5.5 TB/s), not by the segment length (tools/bench_opt.py)
constexpr int kOptRows = 32, kOptCiw = 32;
extern "C" int stedm_adamw_ema_pack_piece(int* rows, int* ciw) {
  STEDM_CHECK_ARG(rows && ciw, "adamw_ema_pack_piece: null pointer");
  *rows = kOptRows; *ciw = kOptCiw;
  return 0;
}

extern "C" int stedm_adamw_ema_pack(const void* descs, int ndesc, int total_blocks, float lr, float beta1, float beta2, float eps, float weight_decay,
                                    int step, float ema_decay, float grad_scale, void* stream) {
  STEDM_CHECK_ARG(descs && ndesc > 0 && total_blocks > 0 && step >= 1, "adamw_ema_pack: bad args");
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  const size_t lds = (size_t)kOptRows * (kOptCiw * 9 + 1) * sizeof(float);
  adamw_ema_pack_kernel<kOptRows, kOptCiw><<<total_blocks, 256, lds, as_stream(stream)>>>((const FusedOptDesc*)descs, ndesc, lr, beta1, beta2, eps, weight_decay,
                                                                                          bc1, sqrtf(bc2), ema_decay, grad_scale, nullptr, nullptr);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_adamw_ema_pack_sched(const void* descs, int ndesc, int total_blocks, float beta1, float beta2, float eps, float weight_decay,
                                          const float* sched, const int* sched_idx, float grad_scale, void* stream) {
  STEDM_CHECK_ARG(descs && ndesc > 0 && total_blocks > 0 && sched && sched_idx, "adamw_ema_pack_sched: bad args");
  const size_t lds = (size_t)kOptRows * (kOptCiw * 9 + 1) * sizeof(float);
  adamw_ema_pack_kernel<kOptRows, kOptCiw><<<total_blocks, 256, lds, as_stream(stream)>>>((const FusedOptDesc*)descs, ndesc, 0.f, beta1, beta2, eps, weight_decay,
                                                                                          1.f, 1.f, 0.f, grad_scale, sched, sched_idx);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_ema_update(const void* table, const int* chunk_tensor, const long* chunk_off, int nchunks, float ema_decay, void* stream) {
  STEDM_CHECK_ARG(table && chunk_tensor && chunk_off && nchunks > 0 && ema_decay >= 0.f && ema_decay <= 1.f, "ema_update: bad args");
  ema_update_kernel<<<nchunks, 256, 0, as_stream(stream)>>>((const OptTensor*)table, chunk_tensor, chunk_off, 1.0f - ema_decay);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ SpatialTransformer backward pieces
// LayerNorm backward over rows of `dim` (BasicTransformerBlock.norm1/2/3, attention.py:205-215): xh = (x - mean) rstd, dxh = dy gamma,
//   dx = add + rstd (dxh - mean(dxh) - xh mean(dxh xh));   dgamma = sum_rows dy xh, dbeta = sum_rows dy.
// One wave per row (a lane holds dim / 64 <= 32 elements); a block of 4 waves walks its rows and leaves ONE partial row of (dgamma | dbeta)
// in part[block][2][dim]; ln_bwd_fold adds the blocks in a fixed order.
namespace {
__global__ void __launch_bounds__(256) ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma, float eps,
                                                     const float* add, float* dx, float* __restrict__ part, long rows, int dim, int rows_per_block) {
  constexpr int NV = 32;
  __shared__ float sp[4][2][2048];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dg[NV], db[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) { dg[j] = 0.f; db[j] = 0.f; }
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (long row = r0 + wave; row < r1; row += 4) {
    const float* px = x + row * dim;
    const float* pd = dy + row * dim;
    float xv[NV], dv[NV];
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int k = lane + 64 * j;
      xv[j] = k < dim ? px[k] : 0.f;
      dv[j] = k < dim ? pd[k] : 0.f;
      s1 += xv[j];
    }
    const float mean = wave_sum(s1) / dim;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) { const float d = lane + 64 * j < dim ? xv[j] - mean : 0.f; s2 += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / dim + eps);
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int k = lane + 64 * j;
      if (k < dim) {
        const float xh = (xv[j] - mean) * rstd;
        const float dxh = dv[j] * gamma[k];
        a1 += dxh; a2 += dxh * xh;
        dg[j] += dv[j] * xh; db[j] += dv[j];
        xv[j] = xh; dv[j] = dxh;
      }
    }
    const float m1 = wave_sum(a1) / dim, m2 = wave_sum(a2) / dim;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int k = lane + 64 * j;
      if (k < dim) {
        float v = rstd * (dv[j] - m1 - xv[j] * m2);
        if (add) v += add[row * dim + k];
        dx[row * dim + k] = v;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int k = lane + 64 * j;
    if (k < dim) { sp[wave][0][k] = dg[j]; sp[wave][1][k] = db[j]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * dim; i += 256) {
    const int w = i / dim, k = i - w * dim;
    part[((long)blockIdx.x * 2 + w) * dim + k] = sp[0][w][k] + sp[1][w][k] + sp[2][w][k] + sp[3][w][k];     // fixed order
  }
}

__global__ void __launch_bounds__(256) ln_bwd_fold_kernel(const float* __restrict__ part, int nblk, int dim, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * dim) return;
  const int w = i / dim, k = i - w * dim;
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += part[((long)b * 2 + w) * dim + k];     // fixed order
  float* d = w == 0 ? dgamma : dbeta;
  d[k] = accumulate ? d[k] + s : s;
}

// GEGLU backward (attention.py:37-44: out = value * gelu(gate), exact erf GELU): g [M][2 I] = (value | gate), dh [M][I] -> dg [M][2 I]
__global__ void __launch_bounds__(256) geglu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ dh, float* __restrict__ dg, long M, int I) {
  const long total = M * I;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / I;
    const int c = (int)(i - row * I);
    const float val = g[row * 2 * I + c], gate = g[row * 2 * I + I + c], d = dh[i];
    const float cdf = 0.5f * (1.0f + erff(gate * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * gate * gate);
    dg[row * 2 * I + c] = d * gate * cdf;
    dg[row * 2 * I + I + c] = d * val * (cdf + gate * pdf);
  }
}
}  // namespace

extern "C" int stedm_ln_bwd_blocks(long rows) { return (int)((rows + 63) / 64 < 2048 ? (rows + 63) / 64 : 2048); }

extern "C" int stedm_ln_bwd(const float* x, const float* dy, const float* gamma, float eps, const float* add, float* dx, float* dgamma, float* dbeta,
                            float* ws, long rows, int dim, int accumulate, void* stream) {
  STEDM_CHECK_ARG(x && dy && gamma && dx && dgamma && dbeta && ws && rows > 0, "ln_bwd: bad args");
  STEDM_CHECK_ARG(dim > 0 && dim <= 2048, "ln_bwd: rows of up to 2048 channels (dim=%d)", dim);
  const int nblk = stedm_ln_bwd_blocks(rows);
  const int rpb = (int)((rows + nblk - 1) / nblk);
  const int nb = (int)((rows + rpb - 1) / rpb);
  ln_bwd_kernel<<<nb, 256, 0, as_stream(stream)>>>(x, dy, gamma, eps, add, dx, ws, rows, dim, rpb);
  STEDM_LAUNCH_CHECK();
  ln_bwd_fold_kernel<<<(2 * dim + 255) / 256, 256, 0, as_stream(stream)>>>(ws, nb, dim, dgamma, dbeta, accumulate);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_geglu_bwd(const float* g, const float* dh, float* dg, long M, int I, void* stream) {
  STEDM_CHECK_ARG(g && dh && dg && M > 0 && I > 0, "geglu_bwd: bad args");
  const long total = M * I;
  const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  geglu_bwd_kernel<<<grid, 256, 0, as_stream(stream)>>>(g, dh, dg, M, I);
  STEDM_LAUNCH_CHECK();
  return 0;
}

