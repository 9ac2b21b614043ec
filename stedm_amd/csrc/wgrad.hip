// Direct weight-gradient kernel of the stride-1 3x3 convolutions (training step):
//   dW[tap][ci][co] = sum over (b, y, x) of X[b][y + ky - 1][x + kx - 1][ci] * dY[b][y][x][co]        (zero padding)
// as nine GEMMs that share both operands: the contraction index is the PIXEL, which is the slow index of the NHWC planes, so the
// MFMA fragments (8 consecutive k per lane) are read from LDS with the transposing read ds_read_b64_tr_b16. A workgroup owns a
// 128 (ci) x 64 (co) block of all nine taps (8 waves, two per SIMD, each 32 x 32 x 9 taps = 144 accumulator registers) and walks the pixels in
// units of 64 (whole rows of one image): per unit the haloed patch of X (zero rows / columns for the padding) and the 64 dY rows
// are brought to LDS once and serve all nine taps — the tap is just a different LDS row per k. Compared with the im2col + GEMM form
// (stedm_im2col_t16 + conv_rs_kernel) no 9x expanded copy of X exists and the L1 traffic per MFMA is ~5x lower.
// K (the units) is split over `ksplit` workgroups per block of dW; partials are summed in a fixed order by stedm_wgrad_to_oihw.
#include <stdlib.h>

#include "conv_common.hpp"

using namespace stedm;

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgradArgs {
  const uint16_t* x16;    // [B][H][W][Cin]
  const uint16_t* dy16;   // [B][H][W][Cout]
  float* part;            // [ksplit][9][Cin][Cout]
  int B, H, W, Cin, Cout;
  int wshift;             // log2(W)
  int upr;                // image rows per unit (64 / W)
  int upi;                // units per image (H / upr)
  int nunits;             // B * upi
  int ksplit, tiles_n;    // tiles_n = Cout / 64
  int NP, PW;             // patch positions (upr + 2) * (W + 2), patch width W + 2
  int dbg;                // timing experiments (STEDM_WGRAD_DBG, diagnostic builds only): 1 no global loads, 2 no MFMA phase, 4 no LDS stores
  int nks;                // k-steps of 16 pixels per 64-pixel unit (= 4). A RUN-TIME bound on purpose: with the literal 4 the compiler restructures
                          // the k loop of the W <= 16 forms and spills 337 - 369 registers (554 instead of 89 us per launch: found by the round-5
                          // bench after the ablation switch in this bound had become a compile-time constant)
  int oihw;               // 1: part is [ksplit][Cout][Cin][9] (the parameter's own OIHW order: with ksplit == 1 it IS the gradient)
};

constexpr int WG_NPMAX = 198;                 // max over W of (64 / W + 2) * (W + 2): W = 64 -> 3 * 66
// LDS images are plain rows with a padded pitch: the 4 rows x 32 B a 16-lane group reads transposed land on banks 16q + 8(g&1) + 2p
// (X: 128 channels = 256 B + 64 B pad) / 48q + ... (dY: 64 channels = 128 B + 64 B pad), distinct for 4 consecutive rows. A linear
// image keeps every read address = per-lane base + compile-time constant (tap and block offsets are ds_read immediates).
constexpr int WG_XS = 320, WG_YS = 192;
constexpr int WG_XBUF = WG_NPMAX * WG_XS;
constexpr int WG_YBUF = 64 * WG_YS;

__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* base, int off_lo, int off_hi) {
  typedef __attribute__((address_space(3))) s16x4* lp;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base + off_lo));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base + off_hi));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <int W, bool OIHW = false>
__global__ void __launch_bounds__(512) wgrad3x3_kernel(WgradArgs a) {
  constexpr int PW = W + 2, WSH = W == 8 ? 3 : (W == 16 ? 4 : (W == 32 ? 5 : 6)), NP = (64 / W + 2) * PW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sX = smem;                   // [2][WG_XBUF]
  unsigned char* sY = smem + 2 * WG_XBUF;     // [2][WG_YBUF]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;   // 8 waves (2 per SIMD): each 32 input channels x 32 output channels x 9 taps = 144 accumulators
  const int ntile = (a.Cin / 128) * a.tiles_n;
  const int tile = blockIdx.x % ntile, kz = blockIdx.x / ntile;
  const int ci0 = (tile / a.tiles_n) * 128, co0 = (tile % a.tiles_n) * 64;
  const int per = (a.nunits + a.ksplit - 1) / a.ksplit;
  const int u0 = kz * per, u1 = min(a.nunits, u0 + per);

  // ---- staging registers: 16-B chunks of the next unit's X patch (<= 9 per thread) and dY rows (2 per thread)
  constexpr int NXI = (NP * 16 + 511) / 512;
  constexpr bool DEEP = NXI <= 4;     // W <= 16: registers for a second staging set -> a unit's loads get two compute phases to land
  uint4 rxA[NXI], ryA[1], rxB[DEEP ? NXI : 1], ryB[1];
  auto load_unit = [&](int u, auto& rx, auto& ry) {
    const int b = u / a.upi, y0 = (u - b * a.upi) * a.upr;
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
      const int i = tid + 512 * j, row = i >> 4, ch = i & 15;
      rx[j] = make_uint4(0, 0, 0, 0);
      if (row < NP) {
        const int pr = row / PW, pc = row - pr * PW;
        const int y = y0 + pr - 1, x = pc - 1;
        if (y >= 0 && y < a.H && x >= 0 && x < W)
          rx[j] = *reinterpret_cast<const uint4*>(a.x16 + (((long)b * a.H + y) * W + x) * a.Cin + ci0 + ch * 8);
      }
    }
#pragma unroll
    for (int j = 0; j < 1; ++j) {
      const int i = tid + 512 * j, k = i >> 3, ch = i & 7;
      const int y = y0 + (k >> WSH), x = k & (W - 1);
      ry[j] = *reinterpret_cast<const uint4*>(a.dy16 + (((long)b * a.H + y) * W + x) * a.Cout + co0 + ch * 8);
    }
  };
  auto store_unit = [&](int buf, auto& rx, auto& ry) {
    unsigned char* dx = sX + buf * WG_XBUF;
    unsigned char* dy = sY + buf * WG_YBUF;
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
      const int i = tid + 512 * j, row = i >> 4, ch = i & 15;
      if (row < NP) *reinterpret_cast<uint4*>(dx + row * WG_XS + ch * 16) = rx[j];
    }
#pragma unroll
    for (int j = 0; j < 1; ++j) {
      const int i = tid + 512 * j, k = i >> 3, ch = i & 7;
      *reinterpret_cast<uint4*>(dy + k * WG_YS + ch * 16) = ry[j];
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // ---- per-lane constants of the transposed reads: group g = lane >> 4 (h = g >> 1 selects k 8h..8h+7, g & 1 the 16-column half),
  //      lane 4q + p of the group addresses row q, columns 4p..4p+3 of the 4 x 16 block
  const int g = lane >> 4, h = g >> 1, q = (lane & 15) >> 2, p = lane & 3;
  const int colA = (wm * 32 + 16 * (g & 1)) * 2 + 8 * p;   // byte offset of this lane's 4 columns inside an X row
  const int colB = (wn * 32 + 16 * (g & 1)) * 2 + 8 * p;   // ... inside a dY row

  auto compute_unit = [&](int buf) {
    const unsigned char* px = sX + buf * WG_XBUF;
    const unsigned char* py = sY + buf * WG_YBUF;
#pragma unroll 1
    for (int s = 0; s < (STEDM_DBG(a.dbg, 2) ? 0 : a.nks); ++s) {
      const int k_lo = 16 * s + 8 * h + q, k_hi = k_lo + 4;            // this lane's block rows (pixels of the unit)
      const bf16x8 bf = tr_pair(py, k_lo * WG_YS + colB, k_hi * WG_YS + colB);
      // patch row of tap (0, 0); tap (ky, kx) adds the compile-time constant (ky * PW + kx) rows
      const int a_lo = ((k_lo >> WSH) * PW + (k_lo & (W - 1))) * WG_XS + colA;
      const int a_hi = ((k_hi >> WSH) * PW + (k_hi & (W - 1))) * WG_XS + colA;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int toff = ((t / 3) * PW + (t % 3)) * WG_XS;
        const bf16x8 af = tr_pair(px, a_lo + toff, a_hi + toff);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[t], 0, 0, 0);
      }
    }
  };
  const bool ld_on = !STEDM_DBG(a.dbg, 1), st_on = !STEDM_DBG(a.dbg, 4);
  if (u0 < u1) { load_unit(u0, rxA, ryA); store_unit(0, rxA, ryA); }
  if constexpr (DEEP) {
    // LDS holds unit u (computed) and u+1; registers hold u+1 / u+2 (sets A / B alternating)
    if (u0 + 1 < u1 && ld_on) load_unit(u0 + 1, rxA, ryA);
    if (u0 + 2 < u1 && ld_on) load_unit(u0 + 2, rxB, ryB);
    __syncthreads();
    for (int u = u0; u < u1; u += 2) {
      compute_unit(0);
      if (u + 1 < u1 && st_on) store_unit(1, rxA, ryA);
      if (u + 3 < u1 && ld_on) load_unit(u + 3, rxA, ryA);
      __syncthreads();
      if (u + 1 >= u1) break;
      compute_unit(1);
      if (u + 2 < u1 && st_on) store_unit(0, rxB, ryB);
      if (u + 4 < u1 && ld_on) load_unit(u + 4, rxB, ryB);
      __syncthreads();
    }
  } else {
    __syncthreads();
    for (int u = u0; u < u1; ++u) {
      const int buf = (u - u0) & 1;
      if (u + 1 < u1 && ld_on) load_unit(u + 1, rxA, ryA);
      compute_unit(buf);
      if (u + 1 < u1 && st_on) store_unit(buf ^ 1, rxA, ryA);
      __syncthreads();
    }
  }

  const int col = lane & 31, hh = lane >> 5;
  if constexpr (OIHW) {
    // ---- partial dW[kz][co][ci][tap], the parameter's own OIHW order: with ksplit == 1 no reduce / transpose pass exists at all, otherwise the
    // reduce is a plain streaming sum. An accumulator's lanes run along co, so direct stores would put every lane on another 4.6-KB row (first
    // version: +14 us per launch). The block goes through LDS instead (the unit buffers are dead): rows [32 co][128 ci x 9 taps] of 4608 contiguous
    // bytes each, two passes (the waves of one co half write, all 512 threads stream the rows out as 16-B stores, fully coalesced). Row pitch
    // 1156 floats: 16-B aligned rows, and the 32 lanes of a write instruction (one row each) fall 2-way on the banks, which costs a
    // ds_write_b32 nothing (MI355X_MICROARCH.md, LDS).
    constexpr int PITCH = 1156;
    static_assert(32 * PITCH * 4 <= 2 * WG_XBUF + 2 * WG_YBUF, "staging rows must fit the unit buffers");
    float* stg = reinterpret_cast<float*>(smem);
    __syncthreads();              // every wave is done with the last unit's images
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      if (wn == pass) {
        float* row = stg + col * PITCH + (wm * 32 + 4 * hh) * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int e = 0; e < 16; ++e) row[((e & 3) + 8 * (e >> 2)) * 9 + t] = acc[t][e];
      }
      __syncthreads();
      float* dst = a.part + (((long)kz * a.Cout + co0 + pass * 32) * a.Cin + ci0) * 9;
      for (int idx = tid; idx < 32 * 288; idx += 512) {
        const int r = idx / 288, c4 = idx - r * 288;
        const float4 v = *reinterpret_cast<const float4*>(stg + r * PITCH + c4 * 4);
        *reinterpret_cast<float4*>(dst + (long)r * a.Cin * 9 + c4 * 4) = v;
      }
      __syncthreads();
    }
  } else
  // ---- partial dW[kz][tap][ci][co]
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float* dst = a.part + (((long)kz * 9 + t) * a.Cin + ci0 + wm * 32) * a.Cout + co0 + wn * 32 + col;
#pragma unroll
    for (int e = 0; e < 16; ++e) dst[(long)((e & 3) + 8 * (e >> 2) + 4 * hh) * a.Cout] = acc[t][e];
  }
}

// out[i] (+)= sum over z < nsplit of part[z * n + i], fixed order: the split-K partials of stedm_wgrad3x3_oihw are already in the gradient's order
__global__ void __launch_bounds__(256) sum_planes_kernel(const float4* __restrict__ part, float4* __restrict__ out, long n4, int nsplit, int accumulate) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 v = accumulate ? out[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 16      // (loads of 16 slices in flight, added in slice order: small filters have few columns and up to 128 slices)
    for (int z = 0; z < nsplit; ++z) {
      const float4 w = part[(long)z * n4 + i];
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    out[i] = v;
  }
}

// ---- the same for a 1x1 convolution (skip_connection, attention qkv / proj_out): dW[ci][co] = sum over pixels p of X[p][ci] * dY[p][co].
// No patch geometry: the planes are flat [P][C] and a unit is 64 consecutive pixels. With one tap the accumulators are small, so a workgroup
// owns 128 (ci) x 128 (co) — every wave 32 x 64 — which halves the re-reads of X against the 128 x 64 block of the 3x3 kernel; the loop is
// bound by the operand stream (32 KB per unit and workgroup for 64 MFMAs), not by the MFMA pipe. Replaces stedm_im2col_t16 (two 16-bit
// transposes through HBM) + the dY^T fragment pack + the 1x1-kind GEMM of the GEMM form.
struct Wgrad1Args {
  const uint16_t* x16;    // [P][Cin]
  const uint16_t* dy16;   // [P][Cout]
  float* part;            // [ksplit][Cin][Cout]
  int Cin, Cout, nunits, ksplit, tiles_n;    // tiles_n = Cout / 128
};

constexpr int W1_S = 320;                 // row pitch of both LDS images: 128 channels = 256 B + 64 B pad (see WG_XS)
constexpr int W1_BUF = 64 * W1_S;

__global__ void __launch_bounds__(512) wgrad1x1_kernel(Wgrad1Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sX = smem;                  // [2][W1_BUF]
  unsigned char* sY = smem + 2 * W1_BUF;     // [2][W1_BUF]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;   // 8 waves: 32 input channels x 64 output channels each
  const int ntile = (a.Cin / 128) * a.tiles_n;
  const int tile = blockIdx.x % ntile, kz = blockIdx.x / ntile;
  const int ci0 = (tile / a.tiles_n) * 128, co0 = (tile % a.tiles_n) * 128;
  const int per = (a.nunits + a.ksplit - 1) / a.ksplit;
  const int u0 = kz * per, u1 = min(a.nunits, u0 + per);

  // staging registers: four 16-B chunks per thread and unit (named, macro-expanded: arrays captured by the lambdas went to scratch)
  uint4 rx0, rx1, ry0, ry1;
  const int srow0 = tid >> 4, srow1 = (tid + 512) >> 4, sch = tid & 15;
#define W1_LOAD(U)                                                                                      \
  {                                                                                                     \
    const long p0_ = (long)(U) * 64 + srow0, p1_ = (long)(U) * 64 + srow1;                              \
    rx0 = *reinterpret_cast<const uint4*>(a.x16 + p0_ * a.Cin + ci0 + sch * 8);                         \
    rx1 = *reinterpret_cast<const uint4*>(a.x16 + p1_ * a.Cin + ci0 + sch * 8);                         \
    ry0 = *reinterpret_cast<const uint4*>(a.dy16 + p0_ * a.Cout + co0 + sch * 8);                       \
    ry1 = *reinterpret_cast<const uint4*>(a.dy16 + p1_ * a.Cout + co0 + sch * 8);                       \
  }
#define W1_STORE(BUF)                                                                                   \
  {                                                                                                     \
    *reinterpret_cast<uint4*>(sX + (BUF) * W1_BUF + srow0 * W1_S + sch * 16) = rx0;                     \
    *reinterpret_cast<uint4*>(sX + (BUF) * W1_BUF + srow1 * W1_S + sch * 16) = rx1;                     \
    *reinterpret_cast<uint4*>(sY + (BUF) * W1_BUF + srow0 * W1_S + sch * 16) = ry0;                     \
    *reinterpret_cast<uint4*>(sY + (BUF) * W1_BUF + srow1 * W1_S + sch * 16) = ry1;                     \
  }
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  // transposed-read constants as in wgrad3x3_kernel: group g = lane >> 4 (h = g >> 1 selects k 8h..8h+7, g & 1 the 16-column half),
  // lane 4q + p of the group addresses row q, columns 4p..4p+3 of the 4 x 16 block
  const int g = lane >> 4, h = g >> 1, q = (lane & 15) >> 2, p = lane & 3;
  const int colA = (wm * 32 + 16 * (g & 1)) * 2 + 8 * p;
  const int colB = (wn * 64 + 16 * (g & 1)) * 2 + 8 * p;
  auto compute_unit = [&](int buf) {
    const unsigned char* px = sX + buf * W1_BUF;
    const unsigned char* py = sY + buf * W1_BUF;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k_lo = 16 * s + 8 * h + q, k_hi = k_lo + 4;
      const bf16x8 af = tr_pair(px, k_lo * W1_S + colA, k_hi * W1_S + colA);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 bf = tr_pair(py, k_lo * W1_S + colB + j * 64, k_hi * W1_S + colB + j * 64);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[j], 0, 0, 0);
      }
    }
  };
  if (u0 < u1) { W1_LOAD(u0) W1_STORE(0) }
  __syncthreads();
  for (int u = u0; u < u1; ++u) {
    const int buf = (u - u0) & 1;
    if (u + 1 < u1) W1_LOAD(u + 1)
    compute_unit(buf);
    if (u + 1 < u1) W1_STORE(buf ^ 1)
    __syncthreads();
  }
#undef W1_LOAD
#undef W1_STORE
  // ---- partial dW[kz][ci][co]
  const int col = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float* dst = a.part + ((long)kz * a.Cin + ci0 + wm * 32) * a.Cout + co0 + wn * 64 + j * 32 + col;
#pragma unroll
    for (int e = 0; e < 16; ++e) dst[(long)((e & 3) + 8 * (e >> 2) + 4 * hh) * a.Cout] = acc[j][e];
  }
}

}  // namespace

// 1 when stedm_wgrad3x3 supports the shape, and the split it would use in *ksplit (for sizing `part`: ksplit * 9 * Cin * Cout floats)
extern "C" int stedm_wgrad3x3_plan(int B, int H, int W, int Cin, int Cout, int* ksplit) {
  if (ksplit) *ksplit = 0;
  if (B <= 0 || H <= 0 || (W != 8 && W != 16 && W != 32 && W != 64) || Cin % 128 != 0 || Cout % 64 != 0 || H % (64 / W) != 0) return 0;
  static int cus = 0;
  if (cus == 0) { cus = stedm_device_cus(); if (cus <= 0) cus = 256; }
  const int tiles = (Cin / 128) * (Cout / 64), nunits = B * (H / (64 / W));
  int ks = cus / tiles;                          // whole slices that fit ONE round of the chip (one workgroup is resident per CU)
  const long wbytes = 9L * Cin * Cout * 4;       // every slice costs a partial of this size in HBM (written here, read by the reduce):
  int cap = (int)((64L << 20) / wbytes);         // <= 64 MB of partials, between 16 and 128 slices (the reduce walks them serially). Small filters
  cap = cap < 16 ? 16 : (cap > 128 ? 128 : cap); // (128 -> 128 at 32 x 32: 2 tiles) were capped at 32 slices = 64 workgroups: 100 us for 19 GFLOP

  if (ks > cap) ks = cap;
  if (ks > nunits / 4) ks = nunits / 4;          // and >= 4 units per slice
  if (ks < 1) ks = 1;
  const int per = (nunits + ks - 1) / ks;
  ks = (nunits + per - 1) / per;                 // no empty slices
  if (ksplit) *ksplit = ks;
  return 1;
}

static int wgrad3x3_launch(const void* x16, const void* dy16, float* part, int B, int H, int W, int Cin, int Cout, int mm_dtype, int oihw, void* stream);

extern "C" int stedm_wgrad3x3(const void* x16, const void* dy16, float* part, int B, int H, int W, int Cin, int Cout, int mm_dtype, void* stream) {
  return wgrad3x3_launch(x16, dy16, part, B, H, W, Cin, Cout, mm_dtype, 0, stream);
}

// The same kernel with its partials in the PARAMETER's order: part = [ksplit][Cout][Cin][3][3]. With ksplit == 1 (stedm_wgrad3x3_plan) `part`
// may be the gradient itself; otherwise stedm_sum_planes adds the slices. Replaces stedm_wgrad3x3 + stedm_wgrad_to_oihw (the partial round
// trip through a transposing reduce) in the training step.
extern "C" int stedm_wgrad3x3_oihw(const void* x16, const void* dy16, float* part, int B, int H, int W, int Cin, int Cout, int mm_dtype, void* stream) {
  return wgrad3x3_launch(x16, dy16, part, B, H, W, Cin, Cout, mm_dtype, 1, stream);
}

extern "C" int stedm_sum_planes(const float* part, float* out, long n, int nsplit, int accumulate, void* stream) {
  STEDM_CHECK_ARG(part && out && n > 0 && n % 4 == 0 && nsplit >= 1, "sum_planes: bad args (n %% 4 == 0)");
  STEDM_CHECK_ARG(((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "sum_planes: pointers must be 16-byte aligned");
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  sum_planes_kernel<<<(unsigned)blocks, 256, 0, as_stream(stream)>>>(reinterpret_cast<const float4*>(part), reinterpret_cast<float4*>(out), n4, nsplit, accumulate);
  STEDM_LAUNCH_CHECK();
  return 0;
}

static int wgrad3x3_launch(const void* x16, const void* dy16, float* part, int B, int H, int W, int Cin, int Cout, int mm_dtype, int oihw, void* stream) {
  STEDM_CHECK_ARG(x16 && dy16 && part, "wgrad3x3: null pointer");
  STEDM_CHECK_ARG(!oihw || (reinterpret_cast<uintptr_t>(part) & 15) == 0, "wgrad3x3_oihw: the output must be 16-byte aligned");
  STEDM_CHECK_ARG(mm_dtype == STEDM_BF16, "wgrad3x3: bf16 operands only (the backward pass's operand format)");
  int ks = 0;
  STEDM_CHECK_ARG(stedm_wgrad3x3_plan(B, H, W, Cin, Cout, &ks) == 1, "wgrad3x3: unsupported shape (W in {8,16,32,64}, H %% (64/W) == 0, Cin %% 128 == 0, Cout %% 64 == 0)");
  STEDM_CHECK_ARG((long)B * H * W * (Cin > Cout ? Cin : Cout) < (1L << 31), "wgrad3x3: tensor too large");
  WgradArgs a;
  a.x16 = (const uint16_t*)x16; a.dy16 = (const uint16_t*)dy16; a.part = part;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.wshift = W == 8 ? 3 : (W == 16 ? 4 : (W == 32 ? 5 : 6));
  a.upr = 64 / W; a.upi = H / a.upr; a.nunits = B * a.upi;
  a.ksplit = ks; a.tiles_n = Cout / 64;
  a.PW = W + 2; a.NP = (a.upr + 2) * a.PW;
  static const int dbg = getenv("STEDM_WGRAD_DBG") ? atoi(getenv("STEDM_WGRAD_DBG")) : 0;
#ifdef STEDM_CONV_DIAG
  a.dbg = dbg;
#else
  a.dbg = 0;
  if (dbg) { set_error("wgrad: STEDM_WGRAD_DBG=%d needs a diagnostic build (-DSTEDM_CONV_DIAG=0); the shipped kernels compile no ablation switch", dbg); return 1; }
#endif
  a.oihw = oihw;
  a.nks = 4;
  const size_t lds = 2 * WG_XBUF + 2 * WG_YBUF;
  static bool attr = false;
  if (!attr) {
    STEDM_HIP_TRY(hipFuncSetAttribute((const void*)wgrad3x3_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    STEDM_HIP_TRY(hipFuncSetAttribute((const void*)wgrad3x3_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    STEDM_HIP_TRY(hipFuncSetAttribute((const void*)wgrad3x3_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    STEDM_HIP_TRY(hipFuncSetAttribute((const void*)wgrad3x3_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    STEDM_HIP_TRY(hipFuncSetAttribute((const void*)wgrad3x3_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    STEDM_HIP_TRY(hipFuncSetAttribute((const void*)wgrad3x3_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    STEDM_HIP_TRY(hipFuncSetAttribute((const void*)wgrad3x3_kernel<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    STEDM_HIP_TRY(hipFuncSetAttribute((const void*)wgrad3x3_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  const int grid = (Cin / 128) * a.tiles_n * ks;
  hipStream_t st = as_stream(stream);
  if (oihw) {
    if (W == 8) wgrad3x3_kernel<8, true><<<grid, 512, lds, st>>>(a);
    else if (W == 16) wgrad3x3_kernel<16, true><<<grid, 512, lds, st>>>(a);
    else if (W == 32) wgrad3x3_kernel<32, true><<<grid, 512, lds, st>>>(a);
    else wgrad3x3_kernel<64, true><<<grid, 512, lds, st>>>(a);
  } else if (W == 8) wgrad3x3_kernel<8><<<grid, 512, lds, st>>>(a);
  else if (W == 16) wgrad3x3_kernel<16><<<grid, 512, lds, st>>>(a);
  else if (W == 32) wgrad3x3_kernel<32><<<grid, 512, lds, st>>>(a);
  else wgrad3x3_kernel<64><<<grid, 512, lds, st>>>(a);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// 1 when stedm_wgrad1x1 supports the shape (P %% 64 == 0, Cin %% 128 == 0, Cout %% 128 == 0), and the split it would use in *ksplit
// (for sizing `part`: ksplit * Cin * Cout floats)
extern "C" int stedm_wgrad1x1_plan(long P, int Cin, int Cout, int* ksplit) {
  if (ksplit) *ksplit = 0;
  if (P <= 0 || P % 64 != 0 || Cin % 128 != 0 || Cout % 128 != 0 || P * (Cin > Cout ? Cin : Cout) >= (1L << 31)) return 0;
  static int cus = 0;
  if (cus == 0) { cus = stedm_device_cus(); if (cus <= 0) cus = 256; }
  const int tiles = (Cin / 128) * (Cout / 128), nunits = (int)(P / 64);
  int ks = (2 * cus + tiles - 1) / tiles;        // two workgroups (80 KB of LDS each) are resident per CU
  if (ks > 32) ks = 32;
  if (ks > nunits / 4) ks = nunits / 4;          // >= 4 units per slice
  if (ks < 1) ks = 1;
  const int per = (nunits + ks - 1) / ks;
  ks = (nunits + per - 1) / per;                 // no empty slices
  if (ksplit) *ksplit = ks;
  return 1;
}

extern "C" int stedm_wgrad1x1(const void* x16, const void* dy16, float* part, long P, int Cin, int Cout, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(x16 && dy16 && part, "wgrad1x1: null pointer");
  STEDM_CHECK_ARG(mm_dtype == STEDM_BF16, "wgrad1x1: bf16 operands only (the backward pass's operand format)");
  int ks = 0;
  STEDM_CHECK_ARG(stedm_wgrad1x1_plan(P, Cin, Cout, &ks) == 1, "wgrad1x1: unsupported shape (P %% 64 == 0, Cin %% 128 == 0, Cout %% 128 == 0)");
  Wgrad1Args a;
  a.x16 = (const uint16_t*)x16; a.dy16 = (const uint16_t*)dy16; a.part = part;
  a.Cin = Cin; a.Cout = Cout; a.nunits = (int)(P / 64); a.ksplit = ks; a.tiles_n = Cout / 128;
  const size_t lds = 4 * W1_BUF;
  static bool attr = false;
  if (!attr) {
    STEDM_HIP_TRY(hipFuncSetAttribute((const void*)wgrad1x1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  wgrad1x1_kernel<<<(Cin / 128) * a.tiles_n * ks, 512, lds, as_stream(stream)>>>(a);
  STEDM_LAUNCH_CHECK();
  return 0;
}
