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

// Fast path (256 % cout == 0, cin <= 8): a thread owns one output channel and keeps its 9*cin weights in registers;
// a block covers 8 image rows, the haloed input patch sits in LDS padded to 8 floats per position (two 16-B
// broadcast reads per tap).
constexpr int CI_ROWS = 2;   // small row blocks: 4+ waves per SIMD hide the LDS latency of the broadcast reads
__global__ void __launch_bounds__(256) conv_in_fast_kernel(const float* __restrict__ x1, int c1, const float* __restrict__ x2,
                                                           int c2, int bmod, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out, int H, int W,
                                                           int cout, int rblocks, float* __restrict__ cs) {
  extern __shared__ __attribute__((aligned(16))) float smi[];   // [CI_ROWS+2][W+2][8], then [256][2] statistics partials, then the filter [cout][9 cin | 1]
  const int cin = c1 + c2;
  const int PW = W + 2;
  const int b = blockIdx.x / rblocks, y0 = (blockIdx.x % rblocks) * CI_ROWS;
  const int b2 = bmod > 0 ? b % bmod : b;
  for (int i = threadIdx.x; i < (CI_ROWS + 2) * PW * 8; i += 256) {
    const int ci = i & 7;
    const int pc = (i >> 3) % PW;
    const int pr = (i >> 3) / PW;
    const int sy = y0 + pr - 1, sxx = pc - 1;
    float v = 0.f;
    if (ci < cin && sy >= 0 && sy < H && sxx >= 0 && sxx < W)
      v = ci < c1 ? x1[(((long)b * c1 + ci) * H + sy) * W + sxx] : x2[(((long)b2 * c2 + (ci - c1)) * H + sy) * W + sxx];
    smi[i] = v;
  }
  // the filter through LDS: one coalesced read of the OIHW array (a thread's own 9 * cin weights are cin * 9 * 4 B apart between lanes:
  // read from global they cost 72 requests of 64 cache lines each per wave, more than the block's arithmetic); row stride 9 * cin is odd
  // or padded to odd, so the per-thread gather below is bank-conflict-free
  float* swt = smi + (CI_ROWS + 2) * PW * 8 + 512;
  const int wrow = 9 * cin, wst = wrow | 1;
  for (int i = threadIdx.x; i < cout * wrow; i += 256) { const int r = i / wrow; swt[r * wst + (i - r * wrow)] = w[i]; }
  const int co = threadIdx.x % cout, pl = threadIdx.x / cout, npl = 256 / cout;
  const float bv = bias ? bias[co] : 0.f;
  __syncthreads();
  float wr[9][8];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) wr[tap][ci] = ci < cin ? swt[co * wst + ci * 9 + tap] : 0.f;
  const int rows = min(CI_ROWS, H - y0);
  float ssum = 0.f, ssq = 0.f;
  for (int pix = pl; pix < rows * W; pix += npl) {
    const int yl = pix / W, x = pix - yl * W;
    float acc = bv;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      const float4* pp = reinterpret_cast<const float4*>(smi + ((yl + dy) * PW + x + dx) * 8);
      const float4 u = pp[0], v = pp[1];
      acc = fmaf(u.x, wr[tap][0], acc); acc = fmaf(u.y, wr[tap][1], acc); acc = fmaf(u.z, wr[tap][2], acc); acc = fmaf(u.w, wr[tap][3], acc);
      acc = fmaf(v.x, wr[tap][4], acc); acc = fmaf(v.y, wr[tap][5], acc); acc = fmaf(v.z, wr[tap][6], acc); acc = fmaf(v.w, wr[tap][7], acc);
    }
    out[(((long)b * H + y0 + yl) * W + x) * cout + co] = acc;
    ssum += acc; ssq += acc * acc;
  }
  if (cs) {   // per-(sample, row block, channel) partials: the pixel lanes of a channel meet in LDS, fixed order
    float* sp = smi + (CI_ROWS + 2) * PW * 8;
    sp[threadIdx.x * 2] = ssum; sp[threadIdx.x * 2 + 1] = ssq;
    __syncthreads();
    if (threadIdx.x < cout) {
      float su = 0.f, sq = 0.f;
      for (int l = 0; l < npl; ++l) { su += sp[(l * cout + threadIdx.x) * 2]; sq += sp[(l * cout + threadIdx.x) * 2 + 1]; }
      float* d = cs + (((long)b * rblocks + blockIdx.x % rblocks) * cout + threadIdx.x) * 2;
      d[0] = su; d[1] = sq;
    }
  }
}

extern "C" int stedm_conv_in(const float* x1, int c1, const float* x2, int c2, int x2_bmod, const float* w,
                             const float* bias, float* out, int B, int H, int W, int cout, float* chan_stats, void* stream) {
  STEDM_CHECK_ARG(x1 && w && out, "conv_in: null pointer");
  STEDM_CHECK_ARG((x2 != nullptr) == (c2 > 0), "conv_in: x2/c2 mismatch");
  const int cin = c1 + c2;
  if (cin <= 8 && cout <= 256 && 256 % cout == 0) {
    const size_t ldsf = ((size_t)(CI_ROWS + 2) * (W + 2) * 8 + 512 + (size_t)cout * ((9 * cin) | 1)) * sizeof(float);   // patch | statistics | filter
    if (ldsf <= 64 * 1024 && (!chan_stats || H % CI_ROWS == 0)) {
      const int rblocks = (H + CI_ROWS - 1) / CI_ROWS;
      conv_in_fast_kernel<<<B * rblocks, 256, ldsf, as_stream(stream)>>>(x1, c1, x2, c2, x2_bmod, w, bias, out, H, W, cout, rblocks, chan_stats);
      STEDM_LAUNCH_CHECK();
      return 0;
    }
  }
  if (chan_stats) return 3;   // no statistics epilogue on the generic path: nothing was launched
  const size_t lds = ((size_t)3 * (W + 2) * cin + (size_t)9 * cin * cout) * sizeof(float);
  STEDM_CHECK_ARG(lds <= 64 * 1024, "conv_in: cin=%d cout=%d W=%d needs %zu B LDS (> 64 KiB)", cin, cout, W, lds);
  conv_in_kernel<<<B * H, 256, lds, as_stream(stream)>>>(x1, c1, x2, c2, x2_bmod, w, bias, out, H, W, cout);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// conv_out: block = (sample, 4 rows x 32 columns of output pixels). GroupNorm mean/rstd are folded from the producer-side channel
// partials (chan_stats[B][ceil(HW/256)][c][2], fixed order) in the block prologue. Channels are walked in chunks of 32: the haloed,
// normalised + activated patch chunk [6][34][32(+4 pad)] sits in LDS; a thread owns one pixel and half of the chunk's channels
// for all output channels, reads 4 channels per 16-B LDS load, and takes the weights through the scalar cache (HWIO layout
// [3][3][c][4 or 8], zero-padded: the index is wave-uniform and contiguous per tap and channel quad, so they arrive in SGPRs by wide scalar loads
// and feed the FMAs directly); the two halves meet in LDS.
constexpr int CO_MAXOUT = 8;
constexpr int CO_TR = 4, CO_TC = 32, CO_CH = 32;
template <int COUTP>
__global__ void __launch_bounds__(256) conv_out_kernel(const float* __restrict__ src, int c, const float* __restrict__ cs, int nslab,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, int groups, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int H, int W,
                                                       int cout, int tr, int tc) {
  extern __shared__ __attribute__((aligned(16))) float smo[];
  constexpr int PR = CO_TR + 2, PC = CO_TC + 2, PST = CO_CH + 4;
  float* sscale = smo;                       // [c]
  float* sshift = sscale + c;                // [c]
  float* sx = sshift + c;                    // [PR][PC][PST]
  float* sred = sx + PR * PC * PST;          // [128][COUTP]
  const int tiles = tr * tc;
  const int b = blockIdx.x / tiles, t = blockIdx.x % tiles;
  const int y0 = (t / tc) * CO_TR, x0 = (t % tc) * CO_TC;
  const int cpg = c / groups;
  const int HW = H * W;
  // The patch chunks are software-pipelined through registers: chunk k + 1 is requested before chunk k is multiplied, and chunk 0 before the
  // statistics are folded (a workgroup used to walk statistics -> chunk load -> products -> next chunk load strictly in sequence: 40 us per
  // launch at any batch size, all of it memory latency — every workgroup of the grid is resident at once)
  constexpr int NPRE = (PR * PC * (CO_CH / 4) + 255) / 256;
  float4 pre[NPRE];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
      const int i = threadIdx.x + j * 256;
      const int q = i % (CO_CH / 4);
      const int pc = (i / (CO_CH / 4)) % PC;
      const int pr = i / ((CO_CH / 4) * PC);
      const int sy = y0 + pr - 1, sxx = x0 + pc - 1;
      pre[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < PR * PC * (CO_CH / 4) && sy >= 0 && sy < H && sxx >= 0 && sxx < W && c0 + q * 4 < c)
        pre[j] = *reinterpret_cast<const float4*>(src + (((long)b * H + sy) * W + sxx) * c + c0 + q * 4);
    }
  };
  auto stash = [&](int c0) {      // GroupNorm + SiLU on the way into LDS; halo / padding positions stay exactly zero
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
      const int i = threadIdx.x + j * 256;
      if (i >= PR * PC * (CO_CH / 4)) continue;
      const int q = i % (CO_CH / 4);
      const int pc = (i / (CO_CH / 4)) % PC;
      const int pr = i / ((CO_CH / 4) * PC);
      const int sy = y0 + pr - 1, sxx = x0 + pc - 1;
      float4 v = pre[j];
      if (sy >= 0 && sy < H && sxx >= 0 && sxx < W && c0 + q * 4 < c) {
        const int cc = c0 + q * 4;
        v.x = silu_f(fmaf(v.x, sscale[cc], sshift[cc])); v.y = silu_f(fmaf(v.y, sscale[cc + 1], sshift[cc + 1]));
        v.z = silu_f(fmaf(v.z, sscale[cc + 2], sshift[cc + 2])); v.w = silu_f(fmaf(v.w, sscale[cc + 3], sshift[cc + 3]));
      }
      *reinterpret_cast<float4*>(sx + (pr * PC + pc) * PST + q * 4) = v;
    }
  };
  fetch(0);
  // statistics: a thread sums the slab partials of ONE channel (independent loads), the group totals are folded from LDS in channel order
  double* dsum = reinterpret_cast<double*>(sx);     // [c][2] (the patch area is not in use yet)
  for (int ch = threadIdx.x; ch < c; ch += 256) {
    double su = 0.0, sq = 0.0;
#pragma unroll 4
    for (int k = 0; k < nslab; ++k) {
      const float2 pp = *reinterpret_cast<const float2*>(cs + (((long)b * nslab + k) * c + ch) * 2);
      su += (double)pp.x;
      sq += (double)pp.y;
    }
    dsum[ch * 2] = su; dsum[ch * 2 + 1] = sq;
  }
  __syncthreads();
  float my_scale[4], my_shift[4];                   // (c <= 1024 on this path: the LDS check of the launcher)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = threadIdx.x + j * 256;
    my_scale[j] = 0.f; my_shift[j] = 0.f;
    if (ch < c) {
      const int g = ch / cpg;
      double su = 0.0, sq = 0.0;
      for (int cc = g * cpg; cc < (g + 1) * cpg; ++cc) { su += dsum[cc * 2]; sq += dsum[cc * 2 + 1]; }      // fixed order
      const double inv_n = 1.0 / ((double)cpg * HW);
      const double mean = su * inv_n;
      double var = sq * inv_n - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const float rstd = (float)(1.0 / sqrt(var + (double)eps));
      my_scale[j] = gamma[ch] * rstd;
      my_shift[j] = beta[ch] - (float)mean * my_scale[j];
    }
  }
  __syncthreads();                                  // every thread is done reading dsum (it aliases the patch area)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = threadIdx.x + j * 256;
    if (ch < c) { sscale[ch] = my_scale[j]; sshift[ch] = my_shift[j]; }
  }
  const int pix = threadIdx.x & 127;
  const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);   // wave-uniform: keeps the weight index scalar
  const int yl = pix / CO_TC, xl = pix % CO_TC;
  float acc[COUTP];
#pragma unroll
  for (int o = 0; o < COUTP; ++o) acc[o] = 0.f;
  for (int c0 = 0; c0 < c; c0 += CO_CH) {
    __syncthreads();                                // scale / shift visible (first chunk); the previous chunk's products are done with sx
    stash(c0);
    if (c0 + CO_CH < c) fetch(c0 + CO_CH);
    __syncthreads();
    const int cb = c0 + half * (CO_CH / 2);          // first channel of this half (uniform)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      const float* px = sx + ((yl + dy) * PC + xl + dx) * PST + half * (CO_CH / 2);
#pragma unroll
      for (int ci = 0; ci < CO_CH / 2; ci += 4) {
        const float4 v = *reinterpret_cast<const float4*>(px + ci);
        const float* wp = w + ((long)tap * c + cb + ci) * COUTP;     // [tap][ci][COUTP], zero-padded: 4 * COUTP contiguous floats
#pragma unroll
        for (int o = 0; o < COUTP; ++o) {
          acc[o] = fmaf(v.x, wp[o], acc[o]); acc[o] = fmaf(v.y, wp[COUTP + o], acc[o]);
          acc[o] = fmaf(v.z, wp[2 * COUTP + o], acc[o]); acc[o] = fmaf(v.w, wp[3 * COUTP + o], acc[o]);
        }
      }
    }
  }
  __syncthreads();
  if (half == 1) {
#pragma unroll
    for (int o = 0; o < COUTP; ++o) sred[pix * COUTP + o] = acc[o];
  }
  __syncthreads();
  if (half == 0) {
    const int y = y0 + yl, x = x0 + xl;
    if (y < H && x < W) {
#pragma unroll
      for (int o = 0; o < COUTP; ++o)
        if (o < cout) out[(((long)b * cout + o) * H + y) * W + x] = acc[o] + sred[pix * COUTP + o] + (bias ? bias[o] : 0.f);
    }
  }
}

extern "C" int stedm_conv_out(const float* src, int c, const float* chan_stats, int nslab, const float* gamma, const float* beta,
                              float eps, int groups, const float* w, const float* bias, float* out, int B, int H, int W,
                              int cout, void* stream) {
  STEDM_CHECK_ARG(src && chan_stats && gamma && beta && w && out && nslab > 0, "conv_out: null pointer / nslab");
  STEDM_CHECK_ARG(c % 32 == 0 && c <= 1024 && cout >= 1 && cout <= CO_MAXOUT && groups > 0 && c % groups == 0,
                  "conv_out: need c %% 32 == 0, c <= 1024, c %% groups == 0 and cout <= %d (c=%d cout=%d)", CO_MAXOUT, c, cout);
  const int coutp = cout <= 4 ? 4 : 8;
  const size_t lds = ((size_t)2 * c + (size_t)(CO_TR + 2) * (CO_TC + 2) * (CO_CH + 4) + 128 * coutp) * sizeof(float);
  STEDM_CHECK_ARG(lds <= 64 * 1024, "conv_out: needs %zu B LDS", lds);
  const int tr = (H + CO_TR - 1) / CO_TR, tc = (W + CO_TC - 1) / CO_TC;
  if (coutp == 4)
    conv_out_kernel<4><<<B * tr * tc, 256, lds, as_stream(stream)>>>(src, c, chan_stats, nslab, gamma, beta, eps, groups, w, bias, out, H, W, cout, tr, tc);
  else
    conv_out_kernel<8><<<B * tr * tc, 256, lds, as_stream(stream)>>>(src, c, chan_stats, nslab, gamma, beta, eps, groups, w, bias, out, H, W, cout, tr, tc);
  STEDM_LAUNCH_CHECK();
  return 0;
}
