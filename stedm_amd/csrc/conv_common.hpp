// Shared definitions of the implicit-GEMM convolution kernels.
#pragma once
#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BN = 128;

struct ConvParams {
  stedm_conv_args a;
  int M, Hout, Wout, HWout, Cin, taps;
  int whole, nsamp, trows;  // tile geometry
  int wsplit;               // 1: image rows are WIDER than the tile (Wout % tile == 0): a tile is a run of one row, PW = tile + 2 (conv_rs only)
  int PRs, PW, NP;          // patch rows per sample, patch cols, patch positions
  int tiles_m, tiles_n;
  unsigned mg_hw, mg_w, mg_pw, mg_prpw;   // ceil(2^32 / d) for d = HWout, Wout, PW, PRs*PW: n / d == umulhi(n, mg) while n * d < 2^32
  int ksplit;      // register-streamed 3x3 kernel: K split over 2..16 blocks per tile when the grid would not fill the chip (0/1: none)
  int stats_done;  // set by a launcher whose epilogue wrote a.chan_stats (otherwise the dispatcher runs gn_chan_stats)
  int gn_done;     // set by a launcher that wrote a.gn_out16 itself (otherwise the dispatcher runs gn_apply16c)
  int gn_tile;     // register-streamed kernel: its tiles hold whole samples x whole groups -> the epilogue writes a.gn_out16 (conv_rs_try)
  int gn_coop;     // ... or 2 .. 4: tiles per sample that exchange their channel sums inside the launch (a.gn_coop, conv_rs.inc) and then do the same
  int xcd_m;       // register-streamed kernel: XCD-aware block -> tile order: 0 plain, else gm in {8, 4, 2} = the M-tiles are dealt over gm XCD groups, the N-tiles over 8 / gm (conv_rs_kernel)
  int npers;       // register-streamed 1x1 kind: N-persistent form (one block per M-tile walks all N-tiles; conv_rs_try)
  unsigned* ovf;   // fp16 operand range guard flag (common.hpp) or nullptr; set by the dispatcher for fp16 launches
  int dbg;  // ablation bits (STEDM_CONV_DBG; DIAGNOSTIC BUILDS ONLY, see STEDM_DBG below): 1 no weight DMA, 2 no patch staging, 4 no MFMA, 8 no LDS frag reads
};

// Run-time ablation switches and phase stamps (STEDM_CONV_DBG) exist only in diagnostic builds (-DSTEDM_CONV_DIAG=<bits>, tools/conv_diag.sh;
// bits = 0 gives the run-time switches alone). In the shipped library STEDM_DBG(...) is the constant 0: no branch, no load of p.dbg, nothing
// between the compiler and its exact s_waitcnt counts in the hot loops (a run-time branch around a load costs it those counts).
#ifdef STEDM_CONV_DIAG
#define STEDM_DBG(WORD, BITS) (((WORD) & (BITS)) != 0)
#else
#define STEDM_DBG(WORD, BITS) (0)
#endif

template <typename T>
struct MM;
template <>
struct MM<_Float16> {
  using V8 = f16x8;
  using V4 = f16x4;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(V8 a, V8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <>
struct MM<__bf16> {
  using V8 = bf16x8;
  using V4 = bf16x4;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(V8 a, V8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};


// zero page read by halo / padding / overhang lanes of the LDS-DMA loaders: must cover one pixel row of channels (2*Cin bytes) plus a
// chunk; 132 KB admits Cin up to 65536 (the wgrad GEMMs of the training step have K = B*H*W "channels")
#define STEDM_ZERO_PAGE_BYTES 135168

namespace stedm {
// Fills the tile geometry of `p` for an M-tile of `bm` output pixels. Returns false (with the error set) when the
// spatial shape cannot be tiled that way.
bool conv_geometry(ConvParams& p, int bm, bool allow_wsplit = false);
// v3 (LDS-DMA operands from 16-bit activation planes); 0 ok, > 0 error (message set).
int conv_launch_dma(ConvParams& p, hipStream_t st, bool dry = false);
}  // namespace stedm
