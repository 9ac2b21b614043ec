// Fused implicit-GEMM convolution for gfx950 (MFMA 32x32x16, 64-wide waves).
//
//   out[m][n] = sum_{tap, ci} A[m][tap][ci] * W[n][tap][ci] + bias[n] (+ emb[b][n]) (+ res[m][n])
//   A[m][tap][ci] = act( scale[b][ci] * src[b][sy][sx][ci] + shift[b][ci] )   (0 outside the image)
//
// GEMM view: M = B*Hout*Wout output pixels, N = cout, K = taps * cin.
// Data layout in HBM: activations NHWC fp32; weights pre-packed [cout][tap][cin] 16-bit (hi and,
// for the split-precision parity mode, lo = w - hi).
//
// Tiling: one 256-thread workgroup (4 waves, 2x2) computes a 128(M) x 128(N) tile; each wave
// owns 64x64 = 2x2 MFMA 32x32 accumulators. K is walked chunk-major: for each chunk of BKC
// input channels the *haloed input patch* of the tile is staged ONCE in LDS — GroupNorm affine
// + SiLU + fp32->16-bit (hi/lo) conversion applied on the way in — and all 9 taps read shifted
// rows of that patch, so the activation is fetched and normalised once, not 9 times. Weight tiles
// [128][BKC] per (chunk, tap) are double-buffered in LDS with register prefetch of the next one.
// LDS rows are padded by 16 B so 16-B fragment reads of 16 consecutive rows hit distinct banks.
//
// The two sources src1|src2 implement the skip concat (openaimodel.py:800) without materialising
// it: a K-chunk comes entirely from one of them (c1 % BKC == 0).
#include <stdlib.h>

#include "conv_common.hpp"
using namespace stedm;

constexpr int BM = 128, NTHREADS = 256;

template <int BKC, int NPASS, typename T>
__global__ void __launch_bounds__(NTHREADS, 2) conv_igemm_kernel(const ConvParams p) {
  using V8 = typename MM<T>::V8;
  using V4 = typename MM<T>::V4;
  constexpr int AST = BKC * 2 + 16;  // bytes per patch position (padded)
  constexpr int BST = BKC * 2 + 16;  // bytes per weight row (padded)
  constexpr int NPL = NPASS == 3 ? 2 : 1;  // operand planes (hi, lo)
  constexpr int QC = BKC / 4;        // float4 quads per position
  constexpr int POSL = NTHREADS / QC;  // positions handled per sweep
  constexpr int PCS = BKC / 8;       // 16-B pieces per weight row
  constexpr int BPT = BN * PCS / NTHREADS;  // weight pieces per thread per plane
  constexpr int KSTEPS = BKC / 16;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                                  // [NPL][NP][AST]
  const int a_plane = p.NP * AST;
  unsigned char* sB = smem + NPL * a_plane;                  // [2][NPL][BN][BST]
  constexpr int b_plane = BN * BST;

  const stedm_conv_args& a = p.a;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  const int tile_n = blockIdx.x % p.tiles_n, tile_m = blockIdx.x / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const bool is1x1 = (a.ks == 1);

  // ---- tile origin in (sample, row) space
  int b0, yo0;
  if (p.whole) {
    b0 = tile_m * p.nsamp;
    yo0 = 0;
  } else {
    b0 = m0 / p.HWout;
    yo0 = (m0 - b0 * p.HWout) / p.Wout;
  }
  int srow0;
  if (a.mode == STEDM_CONV_S1) srow0 = yo0 - 1;
  else if (a.mode == STEDM_CONV_DOWN) srow0 = 2 * yo0 - 1;
  else srow0 = (yo0 - 1) >> 1;

  // ---- per-lane A fragment geometry (two 32-row subtiles per wave)
  int f_s[2], f_yl[2], f_y[2], f_x[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int ml = wm * 64 + mi * 32 + r;
    if (is1x1) {
      f_s[mi] = 0; f_yl[mi] = 0; f_y[mi] = 0; f_x[mi] = ml;
    } else if (p.whole) {
      const int s = ml / p.HWout, rem = ml - s * p.HWout;
      f_s[mi] = s; f_yl[mi] = rem / p.Wout; f_x[mi] = rem - f_yl[mi] * p.Wout; f_y[mi] = f_yl[mi];
    } else {
      f_s[mi] = 0; f_yl[mi] = ml / p.Wout; f_x[mi] = ml - f_yl[mi] * p.Wout; f_y[mi] = yo0 + f_yl[mi];
    }
  }
  auto patch_index = [&](int mi, int dy, int dx) -> int {
    if (is1x1) return f_x[mi];
    int prow, pcol;
    if (a.mode == STEDM_CONV_S1) { prow = f_yl[mi] + dy; pcol = f_x[mi] + dx; }
    else if (a.mode == STEDM_CONV_DOWN) { prow = 2 * f_yl[mi] + dy; pcol = 2 * f_x[mi] + dx; }
    else { prow = ((f_y[mi] + dy - 1) >> 1) - srow0; pcol = ((f_x[mi] + dx - 1) >> 1) + 1; }
    return (f_s[mi] * p.PRs + prow) * p.PW + pcol;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nchunks = p.Cin / BKC;
  const int taps = p.taps;
  const int nsteps = nchunks * taps;

  // ---- weight tile prefetch registers
  uint4 wreg[NPL][BPT];
  unsigned bad_src = 0u;      // fp16 range guard: inf / NaN among the operands this kernel rounds from the fp32 source (common.hpp)
  auto load_w = [&](int step) {
    const int chunk = step / taps, tap = step - chunk * taps;
    const int c0 = chunk * BKC;
#pragma unroll
    for (int j = 0; j < BPT; ++j) {
      const int i = tid + j * NTHREADS;
      const int row = i / PCS, pc = i - row * PCS;
      const int n = n0 + row;
      const long off = ((long)n * taps + tap) * p.Cin + c0 + pc * 8;  // in 16-bit elements
      if (n < a.cout) {
        wreg[0][j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.w_hi) + off);
        if (NPL == 2) wreg[NPL - 1][j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.w_lo) + off);
      } else {
        wreg[0][j] = make_uint4(0, 0, 0, 0);
        if (NPL == 2) wreg[NPL - 1][j] = make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int j = 0; j < BPT; ++j) {
      const int i = tid + j * NTHREADS;
      const int row = i / PCS, pc = i - row * PCS;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
        *reinterpret_cast<uint4*>(sB + (buf * NPL + pl) * b_plane + row * BST + pc * 16) = wreg[pl][j];
    }
  };

  // ---- patch staging (GN affine + act + convert), one chunk of BKC channels
  const int q = tid % QC, pl0 = tid / QC;
  auto load_patch = [&](int chunk) {
    const int c0 = chunk * BKC;
    const float* src;
    int Cs, cs, bmod;
    if (c0 < a.c1) { src = a.src1; Cs = a.c1; cs = c0; bmod = 0; }
    else { src = a.src2; Cs = a.c2; cs = c0 - a.c1; bmod = a.src2_bmod; }
    const bool affine = a.scale != nullptr;
    int cur_b = -1;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    // incremental (s, prow, pcol) decode of position pos = pl0 + it*POSL
    int pcol = pl0, prow = 0, s = 0;
    if (!is1x1) {
      while (pcol >= p.PW) { pcol -= p.PW; if (++prow == p.PRs) { prow = 0; ++s; } }
    }
    for (int pos = pl0; pos < p.NP; pos += POSL) {
      int b;
      long goff;
      bool valid;
      if (is1x1) {
        const int m = m0 + pos;
        valid = m < p.M;
        b = valid ? m / p.HWout : 0;
        const int bs = bmod > 0 ? b % bmod : b;
        goff = ((long)bs * p.HWout + (m - b * p.HWout)) * Cs + cs + q * 4;
      } else {
        b = b0 + s;
        const int sy = srow0 + prow, sx = pcol - 1;
        valid = (b < a.B) && (sy >= 0) && (sy < a.Hin) && (sx >= 0) && (sx < a.Win);
        const int bs = bmod > 0 ? b % bmod : b;
        goff = (((long)bs * a.Hin + sy) * a.Win + sx) * Cs + cs + q * 4;
      }
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) {
        v = *reinterpret_cast<const float4*>(src + goff);
        if (affine) {
          if (b != cur_b) {
            cur_b = b;
            sc = *reinterpret_cast<const float4*>(a.scale + (long)b * p.Cin + c0 + q * 4);
            sh = *reinterpret_cast<const float4*>(a.shift + (long)b * p.Cin + c0 + q * 4);
          }
          v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
          v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        }
        if (a.act == 1) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
      }
      V4 hi;
      hi[0] = (T)v.x; hi[1] = (T)v.y; hi[2] = (T)v.z; hi[3] = (T)v.w;
      *reinterpret_cast<V4*>(sA + pos * AST + q * 8) = hi;
      bad_src |= f16_over_v4<T>(hi);
      if (NPL == 2) {
        V4 lo;
        lo[0] = (T)(v.x - (float)hi[0]); lo[1] = (T)(v.y - (float)hi[1]);
        lo[2] = (T)(v.z - (float)hi[2]); lo[3] = (T)(v.w - (float)hi[3]);
        *reinterpret_cast<V4*>(sA + a_plane + pos * AST + q * 8) = lo;
      }
      if (!is1x1) {
        pcol += POSL;
        while (pcol >= p.PW) { pcol -= p.PW; if (++prow == p.PRs) { prow = 0; ++s; } }
      }
    }
  };

  // ---- main loop
  load_w(0);
  for (int step = 0; step < nsteps; ++step) {
    const int chunk = step / taps, tap = step - chunk * taps;
    const int buf = step & 1;
    if (tap == 0) {
      __syncthreads();  // all reads of the previous patch are done
      load_patch(chunk);
    }
    store_w(buf);       // buffer `buf` was last read two steps ago (barrier of step-1 passed)
    __syncthreads();
    if (step + 1 < nsteps) load_w(step + 1);

    const int dy = is1x1 ? 0 : tap / 3, dx = is1x1 ? 0 : tap - (tap / 3) * 3;
    const unsigned char* pa0 = sA + patch_index(0, dy, dx) * AST + h * 16;
    const unsigned char* pa1 = sA + patch_index(1, dy, dx) * AST + h * 16;
    const unsigned char* pb0 = sB + (buf * NPL) * b_plane + (wn * 64 + r) * BST + h * 16;
    const unsigned char* pb1 = pb0 + 32 * BST;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      V8 ah[2], bh[2];
      ah[0] = *reinterpret_cast<const V8*>(pa0 + ks * 32);
      ah[1] = *reinterpret_cast<const V8*>(pa1 + ks * 32);
      bh[0] = *reinterpret_cast<const V8*>(pb0 + ks * 32);
      bh[1] = *reinterpret_cast<const V8*>(pb1 + ks * 32);
      if (NPASS == 3) {
        V8 al[2], bl[2];
        al[0] = *reinterpret_cast<const V8*>(pa0 + a_plane + ks * 32);
        al[1] = *reinterpret_cast<const V8*>(pa1 + a_plane + ks * 32);
        bl[0] = *reinterpret_cast<const V8*>(pb0 + b_plane + ks * 32);
        bl[1] = *reinterpret_cast<const V8*>(pb1 + b_plane + ks * 32);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = MM<T>::mfma(al[i], bh[j], acc[i][j]);
            acc[i][j] = MM<T>::mfma(ah[i], bl[j], acc[i][j]);
          }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = MM<T>::mfma(ah[i], bh[j], acc[i][j]);
    }
  }

  // ---- epilogue: bias + emb broadcast + residual, NHWC store (lanes along N -> 128-B rows)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + r;
    if (n >= a.cout) continue;
    const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v = acc[i][j][e] + bv;
        if (a.emb) v += a.emb[(long)(m / p.HWout) * a.emb_bstride + n];
        const long o = (long)m * a.cout + n;
        if (a.res) v += a.res[o];
        a.out[o] = v;
      }
    }
  }
  f16_guard_commit(p.ovf, bad_src, STEDM_F16G_CONV_SRC);
}

template <int BKC, int NPASS, typename T>
static int launch(const ConvParams& p, hipStream_t st) {
  constexpr int AST = BKC * 2 + 16, BST = BKC * 2 + 16, NPL = NPASS == 3 ? 2 : 1;
  const size_t lds = (size_t)NPL * p.NP * AST + (size_t)2 * NPL * BN * BST;
  if (lds > 160 * 1024) {
    set_error("conv_igemm: tile needs %zu B of LDS (> 160 KiB): Hin=%d Win=%d mode=%d", lds, p.a.Hin, p.a.Win, p.a.mode);
    return 1;
  }
  auto k = conv_igemm_kernel<BKC, NPASS, T>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("conv_igemm: hipFuncSetAttribute(%zu) failed: %s", lds, hipGetErrorString(e));
      return 2;
    }
  }
  k<<<p.tiles_m * p.tiles_n, NTHREADS, lds, st>>>(p);
  STEDM_LAUNCH_CHECK();
  return 0;
}

bool stedm::conv_geometry(ConvParams& p, int bm, bool allow_wsplit) {
  const stedm_conv_args& a = p.a;
  p.tiles_n = (a.cout + BN - 1) / BN;
  p.tiles_m = (p.M + bm - 1) / bm;
  p.wsplit = 0;
  if (a.ks == 1) {
    p.whole = 0; p.nsamp = 1; p.trows = 0; p.PRs = 1; p.PW = bm; p.NP = bm;
    return true;
  }
  if (p.HWout <= bm) {
    if (bm % p.HWout != 0) { set_error("conv_igemm: Hout*Wout=%d must divide %d", p.HWout, bm); return false; }
    p.whole = 1; p.nsamp = bm / p.HWout; p.trows = p.Hout;
    p.tiles_m = (a.B + p.nsamp - 1) / p.nsamp;
  } else if (allow_wsplit && p.Wout > bm && p.Wout % bm == 0 && (a.mode == STEDM_CONV_S1 || a.mode == STEDM_CONV_UP_SUBPIXEL || a.mode == STEDM_CONV_S2D)) {
    // rows wider than the tile (the 512-pixel rows of the first stage's decoder): a tile is a run of `bm` pixels of ONE row
    p.whole = 0; p.nsamp = 1; p.trows = 1; p.wsplit = 1;
  } else {
    if (bm % p.Wout != 0 || p.HWout % bm != 0) {
      set_error("conv_igemm: unsupported spatial shape %dx%d (need Wout | %d and %d | Hout*Wout)", p.Hout, p.Wout, bm, bm);
      return false;
    }
    p.whole = 0; p.nsamp = 1; p.trows = bm / p.Wout;
  }
  if (a.mode == STEDM_CONV_S1 || a.mode == STEDM_CONV_UP_SUBPIXEL || a.mode == STEDM_CONV_S2D) p.PRs = p.trows + 2;
  else if (a.mode == STEDM_CONV_DOWN) p.PRs = 2 * p.trows + 1;
  else p.PRs = (p.trows + 1) / 2 + 2;
  p.PW = (p.wsplit ? bm : a.Win) + 2;
  p.NP = p.nsamp * p.PRs * p.PW;
  return true;
}

static int conv_dispatch(ConvParams& p, void* stream);
static int conv_setup(ConvParams& p);

// 1 when the tiled 3x3 kernels have a tiling for an Hout x Wout output grid (conv_geometry: runs of 128 or 256 pixels aligned with the image
// rows), 0 otherwise - the caller then takes the im2col + flat GEMM form (stedm_im2col_rows16). No launch, no error message.
extern "C" int stedm_conv3x3_tiles_ok(int Hout, int Wout) {
  if (Hout <= 0 || Wout <= 0) return 0;
  if (Wout > 256 && Wout % 256 == 0) return 1;      // rows wider than a tile (the register-streamed kernel's row-run form)
  for (int bm = 256; bm >= 128; bm >>= 1) {
    const int HW = Hout * Wout;
    if (HW <= bm) { if (bm % HW == 0) return 1; continue; }
    if (bm % Wout == 0 && HW % bm == 0) return 1;
  }
  return 0;
}

extern "C" int stedm_conv_fused_skip_ok(const stedm_conv_args* args) {
  if (!args || !args->src16b_hi || !args->src16_hi) return 0;
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.a = *args;
  if (conv_setup(p) != 0) return 0;
  return conv_launch_dma(p, nullptr, /*dry=*/true) == 0 ? 1 : 0;
}

// Would the register-streamed kernel (fragment-order weights, conv_rs.inc) run this problem? Then w_hi / w_lo are never read and the
// caller may skip packing them (pass any non-NULL w_hi). Same decision path as stedm_conv_igemm, nothing is launched. (npass = 3: with the
// hi + lo streams in w_frag16.)
extern "C" int stedm_conv_rs_ok(const stedm_conv_args* args) {
  if (!args || !args->src16_hi || (!args->w_frag && !args->w_frag16) || (args->npass != 1 && !args->w_frag16)) return 0;
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.a = *args;
  if (conv_setup(p) != 0) return 0;
  return conv_launch_dma(p, nullptr, /*dry=*/true) == 0 ? 1 : 0;
}

extern "C" int stedm_conv_igemm(const stedm_conv_args* args, void* stream) {
  STEDM_CHECK_ARG(args, "conv_igemm: null args");
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.a = *args;
  const stedm_conv_args& a = p.a;
  STEDM_CHECK_ARG(!a.chan_stats || a.out || (a.out16_hi && !a.out16_lo && a.npass == 1), "conv_igemm: chan_stats needs the fp32 output (or, single product, the 16-bit-only form)");
  STEDM_CHECK_ARG(a.out16_stride == 0 || (a.out16_hi && !a.out16_lo && a.npass == 1 && a.out16_stride >= a.cout && a.out16_stride % 4 == 0 && a.cout % 4 == 0 &&
                                          !a.ln_gamma && !a.qkv_q && !a.act_out),
                  "conv_igemm: out16_stride needs out16_hi alone (single product), stride >= cout, both multiples of 4, plain epilogue");
  STEDM_CHECK_ARG(!a.gn_out16 || (a.out && a.chan_stats && a.gn_gamma && a.gn_beta && a.gn_groups > 0 && a.cout % a.gn_groups == 0 &&
                                  (a.npass == 1 || a.gn_out16_lo) && a.mode == STEDM_CONV_S1 && (a.gn_act == 0 || a.gn_act == 1)),
                  "conv_igemm: gn_out16 needs out, chan_stats, gamma / beta, groups dividing cout, stride 1 and a single-product mode (or gn_out16_lo)");
  STEDM_CHECK_ARG(!a.gn_out16_lo || a.gn_out16, "conv_igemm: gn_out16_lo without gn_out16");
  int rc = conv_dispatch(p, stream);
  if (rc != 0) return rc;
  if (a.chan_stats && !p.stats_done && !a.out) {      // (cannot happen: conv_rs_try declines the 16-bit-only form when its epilogue would not write the statistics)
    set_error("conv_igemm: the 16-bit-only form with chan_stats ran on a kernel without the statistics epilogue");
    return 1;
  }
  if (a.chan_stats && !p.stats_done) {
    // the kernel that ran has no statistics epilogue: one extra pass over the output
    const int up = (a.mode == STEDM_CONV_UP || a.mode == STEDM_CONV_UP_SUBPIXEL) ? 4 : 1, down = a.mode == STEDM_CONV_DOWN ? 4 : 1;
    rc = stedm_gn_chan_stats(a.out, a.cout, a.B, a.Hin * a.Win * up / down, a.chan_nslab, a.chan_stats, stream);
    if (rc != 0) return rc;
  }
  if (a.gn_out16 && !p.gn_done) {
    // no pass of this launch owned whole groups: the consumer's GroupNorm as its own pass, from the statistics just written
    const int HW = a.Hin * a.Win;
    rc = stedm_gn_apply16c_mr(a.out, a.cout, a.chan_stats, a.chan_nslab > 0 ? a.chan_nslab : stedm_gn_chan_nslab(HW), nullptr, 0, nullptr, 0, 0, a.gn_gamma,
                              a.gn_beta, a.gn_eps, a.gn_groups, a.gn_act, a.B, HW, a.gn_out16, a.gn_out16_lo, nullptr, nullptr, a.gn_mr, a.mm_dtype, stream);
  }
  return rc;
}

// validates the arguments and fills the derived sizes of `p`
static int conv_setup(ConvParams& p) {
  const stedm_conv_args& a = p.a;
  STEDM_CHECK_ARG((a.src1 || a.src16_hi) && (a.w_hi || (a.mode == STEDM_CONV_S2D && (a.w_frag || (a.npass == 3 && a.w_frag16)))) && (a.out || a.out16_hi || a.qkv_q),
                  "conv_igemm: null src/w_hi/out");
  STEDM_CHECK_ARG(!a.ln_gamma || (a.ln_beta && (a.out || a.out16_hi) && !a.out16_lo && !a.res && !a.emb && !a.chan_stats && !a.gn_out16 && !a.act_out && !a.qkv_q &&
                                  a.src16_hi && !a.src1 && !a.src16b_hi && a.w_frag && a.ks == 1 && a.mode == STEDM_CONV_S1 && a.npass == 1 && a.cout <= 128 &&
                                  a.cout % 4 == 0),
                  "conv_igemm: the LayerNorm epilogue needs a 1x1 GEMM with cout <= 128 (a multiple of 4), src16 + w_frag, a single-product mode and no other epilogue extra");
  STEDM_CHECK_ARG(!a.qkv_q || (a.qkv_k && a.qkv_vt && !a.out && !a.out16_hi && !a.res && !a.emb && !a.chan_stats && !a.gn_out16 && !a.act_out && a.src16_hi && !a.src1 &&
                               !a.src16b_hi && a.w_frag && a.ks == 1 && a.mode == STEDM_CONV_S1 && a.npass == 1 && a.B == 1 && a.Hin == 1 && a.qkv_heads > 0 &&
                               a.qkv_heads % 2 == 0 && a.cout == 3 * a.qkv_heads * 64 && a.qkv_T > 0 && a.qkv_T % 2 == 0 && a.qkv_Tp >= a.qkv_T &&
                               a.qkv_Tp % 2 == 0 && a.Win % a.qkv_T == 0),
                  "conv_igemm: the qkv epilogue needs a flat 1x1 GEMM (B = Hin = 1, Win = nb * T rows), cout = 3 * heads * 64 with an even head count, even T / Tp, "
                  "src16 + w_frag, a single-product mode and no other output or epilogue extra");
  STEDM_CHECK_ARG(a.pad_br == 0 || a.mode == STEDM_CONV_S2D, "conv_igemm: pad_br belongs to the space-to-depth form");
  STEDM_CHECK_ARG(a.mode != STEDM_CONV_S2D || (a.src16_hi && !a.src1 && a.ks == 3 && !a.src16b_hi &&
                                                ((a.npass == 1 && a.w_frag) || (a.npass == 3 && a.w_frag16 && a.src16_lo))),
                  "conv_igemm: the space-to-depth form needs src16 planes, ks=3 and w_frag (single product) or w_frag16 + src16_lo (3 products)");
  STEDM_CHECK_ARG((!a.act_out && !a.out16_hi) || (a.src16_hi && !a.src1), "conv_igemm: act_out/out16 need the DMA path (src16 only)");
  STEDM_CHECK_ARG(!a.src1 || (a.src2 != nullptr) == (a.c2 > 0), "conv_igemm: src2/c2 mismatch");
  STEDM_CHECK_ARG(!a.src16_hi || a.npass == 1 || a.src16_lo, "conv_igemm: npass=3 needs src16_lo");
  STEDM_CHECK_ARG(a.ks == 1 || a.ks == 3, "conv_igemm: ks must be 1 or 3");
  STEDM_CHECK_ARG(a.mode >= 0 && a.mode <= 4, "conv_igemm: bad mode %d", a.mode);
  STEDM_CHECK_ARG(a.mode != STEDM_CONV_UP_SUBPIXEL || (a.src16_hi && !a.src1 && a.ks == 3), "conv_igemm: sub-pixel upsample needs the DMA path (src16) and ks=3");
  STEDM_CHECK_ARG(a.ks == 3 || a.mode == STEDM_CONV_S1, "conv_igemm: 1x1 supports stride 1 only");
  STEDM_CHECK_ARG(a.npass == 1 || a.npass == 3, "conv_igemm: npass must be 1 or 3");
  STEDM_CHECK_ARG(a.npass == 1 || a.w_lo || (a.mode == STEDM_CONV_S2D && a.w_frag16), "conv_igemm: npass=3 needs w_lo");
  STEDM_CHECK_ARG((a.scale != nullptr) == (a.shift != nullptr), "conv_igemm: scale/shift must come together");
  STEDM_CHECK_ARG(a.B > 0 && a.Hin > 0 && a.Win > 0 && a.cout > 0, "conv_igemm: bad sizes");
  STEDM_CHECK_ARG(a.mm_dtype == STEDM_F16 || a.mm_dtype == STEDM_BF16, "conv_igemm: bad mm_dtype %d", a.mm_dtype);
  p.Cin = a.c1 + a.c2;
  p.taps = (a.mode == STEDM_CONV_UP_SUBPIXEL || a.mode == STEDM_CONV_S2D) ? 4 : a.ks * a.ks;
  if (a.mode == STEDM_CONV_UP_SUBPIXEL || a.mode == STEDM_CONV_S2D) {
    p.Hout = a.Hin; p.Wout = a.Win;     // tiles are cut on the LOW-RES grid; each tile is computed for the 4 output parities
  } else if (a.mode == STEDM_CONV_DOWN) {
    STEDM_CHECK_ARG(a.Hin % 2 == 0 && a.Win % 2 == 0, "conv_igemm: stride-2 needs even Hin/Win");
    p.Hout = a.Hin / 2; p.Wout = a.Win / 2;
  } else if (a.mode == STEDM_CONV_UP) {
    p.Hout = a.Hin * 2; p.Wout = a.Win * 2;
  } else {
    p.Hout = a.Hin; p.Wout = a.Win;
  }
  p.HWout = p.Hout * p.Wout;
  p.M = a.B * p.HWout;
  STEDM_CHECK_ARG(p.Cin % 32 == 0 && (a.c2 == 0 || a.c1 % 32 == 0),
                  "conv_igemm: channel counts must be multiples of 32 (c1=%d c2=%d)", a.c1, a.c2);
  STEDM_CHECK_ARG(!a.src16b_hi || (a.src16_hi && !a.src1 && ((a.w_frag && a.w_frag_b) || (a.w_frag16 && a.w_frag_b16)) && a.ks == 3 && a.mode == STEDM_CONV_S1 && a.npass == 1 && !a.res &&
                                   a.cb > 0 && a.cb % 64 == 0),
                  "conv_igemm: the fused skip phase needs the DMA path (src16 only), 3x3 stride 1, single product, w_frag + w_frag_b, cb %% 64 == 0, no res");
  return 0;
}

static int conv_dispatch(ConvParams& p, void* stream) {
  const stedm_conv_args& a = p.a;
  { const int rc = conv_setup(p); if (rc) return rc; }
  hipStream_t st = as_stream(stream);
  static const int dbg0 = getenv("STEDM_CONV_DBG") ? atoi(getenv("STEDM_CONV_DBG")) : 0;
#ifdef STEDM_CONV_DIAG
  p.dbg = dbg0;
#else
  p.dbg = 0;
  if (dbg0) { set_error("conv_igemm: STEDM_CONV_DBG=%d needs a diagnostic build (tools/conv_diag.sh, STEDM_HIP_LIB); the shipped kernels compile no ablation switch", dbg0); return 1; }
#endif
  p.ovf = a.mm_dtype == STEDM_F16 ? f16_guard_flag() : nullptr;
  if (a.src16_hi) {   // v3: both operands by LDS-DMA from pre-normalised 16-bit planes
    const int rc = conv_launch_dma(p, st);
    if (rc == 0 || !a.src1) return rc;
    // otherwise fall through to the fused fp32-source kernels
  }
  // fp32-source form (GroupNorm folded into the patch loader): the parity mode's stride-2 convs and the opt-in "fused" path
  const int bkc = (a.npass == 1 && p.Cin % 64 == 0 && (a.c2 == 0 || a.c1 % 64 == 0)) ? 64 : 32;
  if (!conv_geometry(p, BM)) return 1;
  const bool f16 = a.mm_dtype == STEDM_F16;
  if (a.npass == 3) return f16 ? launch<32, 3, _Float16>(p, st) : launch<32, 3, __bf16>(p, st);
  if (bkc == 64) return f16 ? launch<64, 1, _Float16>(p, st) : launch<64, 1, __bf16>(p, st);
  return f16 ? launch<32, 1, _Float16>(p, st) : launch<32, 1, __bf16>(p, st);
}
