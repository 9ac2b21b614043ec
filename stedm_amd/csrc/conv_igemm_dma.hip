// v3 implicit-GEMM convolution: both operands arrive by LDS-DMA, the kernel is MFMA + data movement only.
//
//   A operand : 16-bit NHWC activation planes [B][Hin][Win][Cin] written once by stedm_gn_apply16 (GroupNorm affine +
//               SiLU + conversion already applied, skip-concat materialised). For each chunk of BKC channels the haloed
//               patch of the tile is DMA'd into LDS (global_load_lds_dwordx4, per-lane source address = gather; halo /
//               padding lanes read a zero page) and all 9 taps read shifted rows of it.
//   B operand : packed weights [cout][tap][Cin] 16-bit, one [128][BKC] tile per (chunk, tap), DMA'd NBUF-1 steps ahead.
//   LDS images are unpadded (a DMA instruction writes 1 KiB linearly) and XOR-swizzled through the SOURCE address:
//   16-B piece `c` of row `r` lives at piece c ^ ((r / RPB) % CPP), so the 16 lanes of a ds_read_b128 group hit 16
//   distinct 16-B slots whenever their rows are distinct mod 16.
//
//   512 threads: waves 0-3 compute (one per SIMD, (WM*32) x 64 accumulator slab each, fragment reads software-pipelined
//   one k-step ahead with inline-asm ds_read + counted lgkmcnt), waves 4-7 issue the DMA (a handful of instructions per
//   step) and retire it with counted vmcnt; one s_barrier per tap.
#include "conv_common.hpp"
using namespace stedm;

#define GLDS16(gptr, lptr)                                                                                  \
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr),                   \
                                   (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[16384];   // source of halo / padding lanes

constexpr int DMA_MAX_SLOTS = 20;

template <int BKC, int NPASS, typename T, int WM, int NBUF>
__global__ void __launch_bounds__(512, 2) conv_dma_kernel(const ConvParams p) {
  using V8 = typename MM<T>::V8;
  constexpr int BMT = 2 * WM * 32;
  constexpr int BROW = BKC * 2;              // bytes per LDS row (A position / B weight row), unpadded
  constexpr int NPL = NPASS == 3 ? 2 : 1;
  constexpr int CPP = BKC / 8;               // 16-B pieces per row
  constexpr int RPB = 16 / CPP;              // rows per 256-B bank row
  constexpr int PPI = 64 / CPP;              // rows written by one DMA instruction
  constexpr int NBI = BN / PPI;              // DMA instructions per weight plane
  constexpr int NBG = NBI / 4 * NPL;         // weight DMA instructions per loader wave per step
  constexpr int KSTEPS = BKC / 16;
  constexpr int b_plane = BN * BROW;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nai = (p.NP + PPI - 1) / PPI;    // A DMA instructions per plane per chunk
  const int a_plane = nai * 1024;
  const int a_buf = NPL * a_plane;
  unsigned char* sA = smem;                  // [2][NPL][nai*1024]
  unsigned char* sB = smem + 2 * a_buf;      // [NBUF][NPL][BN][BROW]
  int* sIdx = reinterpret_cast<int*>(sB + NBUF * NPL * b_plane);   // [taps][BMT]: row byte offset | swizzle key << 24
  int* sM = sIdx + p.taps * BMT;             // [BMT] global output pixel index m (-1: masked)
  int* sMb = sM + BMT;                       // [BMT] sample index of that pixel
  int* sPix = sMb + BMT;                     // [nai*PPI] source element offset of the patch position (-1: zero page)

  const stedm_conv_args& a = p.a;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = wave >= 4;
  const int tile_n = blockIdx.x % p.tiles_n, tile_m = blockIdx.x / p.tiles_n;
  const int m0 = tile_m * BMT, n0 = tile_n * BN;
  const bool is1x1 = (a.ks == 1);
  const int taps = p.taps;
  const int nchunks = p.Cin / BKC;
  const int nsteps = nchunks * taps;

  int b0, yo0;
  if (p.whole) { b0 = tile_m * p.nsamp; yo0 = 0; }
  else { b0 = m0 / p.HWout; yo0 = (m0 - b0 * p.HWout) / p.Wout; }
  int srow0;
  if (a.mode == STEDM_CONV_S1) srow0 = yo0 - 1;
  else if (a.mode == STEDM_CONV_DOWN) srow0 = 2 * yo0 - 1;
  else srow0 = (yo0 - 1) >> 1;

  // ---- tables (all threads, once): every div/mod of the tile geometry happens here
  for (int i = tid; i < taps * BMT; i += 512) {
    const int tap = i / BMT, ml = i - tap * BMT;
    int pos;
    if (is1x1) pos = ml;
    else {
      int s_, yl, x, y;
      if (p.whole) { s_ = ml / p.HWout; const int rem = ml - s_ * p.HWout; yl = rem / p.Wout; x = rem - yl * p.Wout; y = yl; }
      else { s_ = 0; yl = ml / p.Wout; x = ml - yl * p.Wout; y = yo0 + yl; }
      const int dy = tap / 3, dx = tap - dy * 3;
      int prow, pcol;
      if (a.mode == STEDM_CONV_S1) { prow = yl + dy; pcol = x + dx; }
      else if (a.mode == STEDM_CONV_DOWN) { prow = 2 * yl + dy; pcol = 2 * x + dx; }
      else { prow = ((y + dy - 1) >> 1) - srow0; pcol = ((x + dx - 1) >> 1) + 1; }
      pos = (s_ * p.PRs + prow) * p.PW + pcol;
    }
    sIdx[i] = (pos * BROW) | (((pos / RPB) % CPP) << 24);
  }
  for (int ml = tid; ml < BMT; ml += 512) {
    int m, b;
    if (is1x1 || !p.whole) { m = m0 + ml; b = m / p.HWout; }
    else { const int s_ = ml / p.HWout; b = b0 + s_; m = b * p.HWout + (ml - s_ * p.HWout); }
    sM[ml] = m < p.M ? m : -1;
    sMb[ml] = b;
  }
  for (int pos = tid; pos < nai * PPI; pos += 512) {
    int off = -1;
    if (pos < p.NP) {
      if (is1x1) {
        const int m = m0 + pos;
        if (m < p.M) off = m;
      } else {
        const int s_ = pos / (p.PRs * p.PW), rem = pos - s_ * (p.PRs * p.PW);
        const int prow = rem / p.PW, pcol = rem - prow * p.PW;
        const int b = b0 + s_, sy = srow0 + prow, sx = pcol - 1;
        if (b < a.B && sy >= 0 && sy < a.Hin && sx >= 0 && sx < a.Win) off = (b * a.Hin + sy) * a.Win + sx;
      }
    }
    sPix[pos] = off;   // pixel index; multiplied by Cin below
  }
  __syncthreads();

  if (is_loader) {
    // =============================================================================== LOADER WAVES (DMA only)
    const int lw = wave - 4;
    const uint16_t* a_hi = reinterpret_cast<const uint16_t*>(a.src16_hi);
    const uint16_t* a_lo = reinterpret_cast<const uint16_t*>(a.src16_lo);
    const uint16_t* zero16 = reinterpret_cast<const uint16_t*>(g_zero_page);
    const int nslots = (nai + 3 - lw) / 4;          // this wave's A instructions: g = lw + 4k
    // per-slot, per-lane source element offset (chunk 0); -1: zero page
    int aoff[DMA_MAX_SLOTS];
    const int lpos = lane / CPP, lphys = lane % CPP;
#pragma unroll
    for (int k = 0; k < DMA_MAX_SLOTS; ++k) {
      aoff[k] = -1;
      if (k < nslots) {
        const int pos = (lw + 4 * k) * PPI + lpos;
        const int logical = lphys ^ ((pos / RPB) % CPP);
        const int pix = sPix[pos];
        aoff[k] = pix >= 0 ? pix * p.Cin + logical * 8 : -(logical * 8) - 1;   // negative: zero page, piece encoded
      }
    }
    // weight rows
    long wrow[NBI / 4];
#pragma unroll
    for (int jj = 0; jj < NBI / 4; ++jj) {
      const int piece = (lw + jj * 4) * 64 + lane;
      const int row = piece / CPP, phys = piece % CPP;
      const int logical = phys ^ ((row / RPB) % CPP);
      int n = n0 + row;
      n = n < a.cout ? n : a.cout - 1;
      wrow[jj] = (long)n * taps * p.Cin + logical * 8;
    }
    auto issue_b = [&](int step) {
      const int chunk = step / taps, tap = step - chunk * taps;
      const long so = (long)tap * p.Cin + chunk * BKC;
      unsigned char* dstb = sB + ((step % NBUF) * NPL) * b_plane;
#pragma unroll
      for (int jj = 0; jj < NBI / 4; ++jj) {
        const int j = lw + jj * 4;
        GLDS16(reinterpret_cast<const uint16_t*>(a.w_hi) + wrow[jj] + so, dstb + j * 1024);
        if (NPL == 2) GLDS16(reinterpret_cast<const uint16_t*>(a.w_lo) + wrow[jj] + so, dstb + b_plane + j * 1024);
      }
    };
    // issue this wave's A slots k = kfirst, kfirst + kstride, ... of `chunk`; returns the number of DMA instructions
    auto issue_a = [&](int chunk, int kfirst, int kstride) -> int {
      const int c0 = chunk * BKC;
      unsigned char* dsta = sA + (chunk & 1) * a_buf;
      int cnt = 0;
#pragma unroll
      for (int k = 0; k < DMA_MAX_SLOTS; ++k) {
        if (k < nslots && k >= kfirst && (k - kfirst) % kstride == 0) {
          const int o = aoff[k];
          const uint16_t* sh = o >= 0 ? a_hi + o + c0 : zero16 + (-o - 1) + c0;
          GLDS16(sh, dsta + (lw + 4 * k) * 1024);
          if (NPL == 2) {
            const uint16_t* sl = o >= 0 ? a_lo + o + c0 : zero16 + (-o - 1) + c0;
            GLDS16(sl, dsta + a_plane + (lw + 4 * k) * 1024);
          }
          cnt += NPL;
        }
      }
      return cnt;
    };
#define WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
    auto wait_keep = [&](int keep) {   // all but the `keep` youngest vector-memory ops of this wave have completed
      switch (keep) {
        case 0: WAIT_VM(0); break; case 1: WAIT_VM(1); break; case 2: WAIT_VM(2); break; case 3: WAIT_VM(3); break;
        case 4: WAIT_VM(4); break; case 5: WAIT_VM(5); break; case 6: WAIT_VM(6); break; case 7: WAIT_VM(7); break;
        case 8: WAIT_VM(8); break; case 9: WAIT_VM(9); break; case 10: WAIT_VM(10); break; case 11: WAIT_VM(11); break;
        case 12: WAIT_VM(12); break; case 13: WAIT_VM(13); break; case 14: WAIT_VM(14); break; case 15: WAIT_VM(15); break;
        default: WAIT_VM(16); break;
      }
    };

    // prologue: patch of chunk 0 and the first weight tile(s)
    if (!(p.dbg & 2)) issue_a(0, 0, 1);
    issue_b(0);
    if (NBUF == 3 && nsteps > 1) issue_b(1);
    WAIT_VM(0);
    __builtin_amdgcn_s_barrier();

    const int ta = taps > 1 ? taps - 1 : 1;     // taps over which the next chunk's patch DMA is spread
    for (int step = 0; step < nsteps; ++step) {
      const int chunk = step / taps, tap = step - chunk * taps;
      const int bstep = step + NBUF - 1;
      const bool issued_b = bstep < nsteps && !(p.dbg & 1);
      int keep;
      if (taps == 1) {
        // 1x1: the next chunk's patch must land within this step -> issue it first, keep only the weight DMA
        if (chunk + 1 < nchunks && !(p.dbg & 2)) issue_a(chunk + 1, 0, 1);
        if (issued_b) issue_b(bstep);
        keep = (NBUF == 3 && issued_b) ? NBG : 0;
      } else {
        int na = 0;
        if (issued_b) issue_b(bstep);
        if (chunk + 1 < nchunks && tap < ta && !(p.dbg & 2)) na = issue_a(chunk + 1, tap, ta);
        keep = na + ((NBUF == 3 && issued_b) ? NBG : 0);
      }
      wait_keep(keep);
      __builtin_amdgcn_s_barrier();
    }
#undef WAIT_VM
    return;
  }

  // ================================================================================= COMPUTE WAVES
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  int brow_off[2], bswz[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wn * 64 + j * 32 + r;
    brow_off[j] = row * BROW;
    bswz[j] = (row / RPB) % CPP;
  }
  const int* myIdx = sIdx + wm * (WM * 32) + r;

  f32x16 acc[WM][2];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();   // prologue DMA landed
  for (int step = 0; step < nsteps; ++step) {
    const int chunk = step / taps, tap = step - chunk * taps;
    const unsigned Abase = (unsigned)(uintptr_t)(sA + (chunk & 1) * a_buf);
    const unsigned Bbase = (unsigned)(uintptr_t)(sB + ((step % NBUF) * NPL) * b_plane);
    unsigned pa[WM], pk[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int e = myIdx[tap * BMT + i * 32];
      pa[i] = Abase + (e & 0xFFFFFF);
      pk[i] = (((unsigned)e >> 24) ^ h) << 4;     // (key ^ h) * 16; the k-step contributes bits 5.. via xor below
    }
    V8 ah[2][WM], bh[2][2], al[2][WM], bl[2][2];
#define LDSR(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
#define LOAD_FRAGS(KS, ST)                                                                             \
    {                                                                                                  \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                  \
        const unsigned ba = Bbase + brow_off[j] + ((((KS) * 2 + h) ^ bswz[j]) * 16);                   \
        LDSR(bh[ST][j], ba);                                                                           \
        if (NPASS == 3) LDSR(bl[ST][j], ba + b_plane);                                                 \
      }                                                                                                \
      _Pragma("unroll") for (int i = 0; i < WM; ++i) {                                                 \
        const unsigned aa = pa[i] + (pk[i] ^ ((KS) << 5));                                             \
        LDSR(ah[ST][i], aa);                                                                           \
        if (NPASS == 3) LDSR(al[ST][i], aa + a_plane);                                                 \
      }                                                                                                \
    }
    constexpr int NRD = (WM + 2) * NPL;
    if (p.dbg & 8) { __builtin_amdgcn_s_barrier(); continue; }
    LOAD_FRAGS(0, 0);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int st = ks & 1;
      if (ks + 1 < KSTEPS) {
        if (st == 0) LOAD_FRAGS(ks + 1, 1) else LOAD_FRAGS(ks + 1, 0)
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NRD) : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      if (p.dbg & 4) continue;
      if (NPASS == 3) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = MM<T>::mfma(al[st][i], bh[st][j], acc[i][j]);
            acc[i][j] = MM<T>::mfma(ah[st][i], bl[st][j], acc[i][j]);
          }
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = MM<T>::mfma(ah[st][i], bh[st][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef LOAD_FRAGS
#undef LDSR
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue: bias + emb broadcast + residual, NHWC store (row -> pixel through the LDS table)
  const int rowbase = wm * (WM * 32);
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ml = rowbase + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      const int m = sM[ml];
      if (m < 0) continue;
      const int bq = sMb[ml];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + r;
        if (n >= a.cout) continue;
        float v = acc[i][j][e] + (a.bias ? a.bias[n] : 0.f);
        if (a.emb) v += a.emb[(long)bq * a.emb_bstride + n];
        const long o = (long)m * a.cout + n;
        if (a.res) v += a.res[o];
        a.out[o] = v;
      }
    }
  }
}

template <int BKC, int NPASS, typename T, int WM, int NBUF>
static size_t dma_lds_bytes(const ConvParams& p) {
  constexpr int NPL = NPASS == 3 ? 2 : 1, PPI = 64 / (BKC / 8), BMT = 2 * WM * 32;
  const size_t nai = (p.NP + PPI - 1) / PPI;
  return (size_t)2 * NPL * nai * 1024 + (size_t)NBUF * NPL * BN * BKC * 2 +
         ((size_t)p.taps * BMT + 2 * BMT + nai * PPI) * sizeof(int);
}

template <int BKC, int NPASS, typename T, int WM, int NBUF>
static int dma_launch(const ConvParams& p, hipStream_t st) {
  const size_t lds = dma_lds_bytes<BKC, NPASS, T, WM, NBUF>(p);
  auto k = conv_dma_kernel<BKC, NPASS, T, WM, NBUF>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("conv_igemm(dma): hipFuncSetAttribute(%zu) failed: %s", lds, hipGetErrorString(e));
      return 2;
    }
  }
  k<<<p.tiles_m * p.tiles_n, 512, lds, st>>>(p);
  STEDM_LAUNCH_CHECK();
  return 0;
}

template <int BKC, int NPASS, typename T, int WM>
static int dma_try(ConvParams& p, const ConvParams& q, hipStream_t st) {
  constexpr size_t LDS_MAX = 160 * 1024;
  constexpr int PPI = 64 / (BKC / 8);
  if (((q.NP + PPI - 1) / PPI + 3) / 4 > DMA_MAX_SLOTS) return -1;
  if (dma_lds_bytes<BKC, NPASS, T, WM, 3>(q) <= LDS_MAX) { p = q; return dma_launch<BKC, NPASS, T, WM, 3>(p, st); }
  if (dma_lds_bytes<BKC, NPASS, T, WM, 2>(q) <= LDS_MAX) { p = q; return dma_launch<BKC, NPASS, T, WM, 2>(p, st); }
  return -1;
}

template <int NPASS, typename T>
static int dma_pick(ConvParams& p, hipStream_t st) {
  static int cus = 0;
  if (cus == 0) { cus = stedm_device_cus(); if (cus <= 0) cus = 256; }
  const bool c64 = (NPASS == 1) && p.Cin % 64 == 0;
  for (int wm4 = 1; wm4 >= 0; --wm4) {
    const int bm = wm4 ? 256 : 128;
    ConvParams q = p;
    if (!conv_geometry(q, bm)) continue;
    if (wm4 && (long)q.tiles_m * q.tiles_n * 4 < (long)cus * 3) continue;
    int rc = -1;
    if (NPASS == 1 && c64) rc = wm4 ? dma_try<64, 1, T, 4>(p, q, st) : dma_try<64, 1, T, 2>(p, q, st);
    if (rc >= 0) return rc;
    rc = wm4 ? dma_try<32, NPASS, T, 4>(p, q, st) : dma_try<32, NPASS, T, 2>(p, q, st);
    if (rc >= 0) return rc;
  }
  return -1;
}

int stedm::conv_launch_dma(ConvParams& p, hipStream_t st) {
  const stedm_conv_args& a = p.a;
  if ((long)a.B * a.Hin * a.Win * p.Cin >= (1L << 31)) {
    set_error("conv_igemm(dma): activation tensor too large for 32-bit element offsets");
    return 1;
  }
  if (p.Cin * 2 + 256 > (int)sizeof(g_zero_page)) { set_error("conv_igemm(dma): Cin too large for the zero page"); return 1; }
  const bool f16 = a.mm_dtype == STEDM_F16;
  int rc = a.npass == 3 ? (f16 ? dma_pick<3, _Float16>(p, st) : dma_pick<3, __bf16>(p, st))
                        : (f16 ? dma_pick<1, _Float16>(p, st) : dma_pick<1, __bf16>(p, st));
  if (rc < 0) { set_error("conv_igemm(dma): no tile configuration fits (Hin=%d Win=%d mode=%d Cin=%d)", a.Hin, a.Win, a.mode, p.Cin); return 1; }
  return rc;
}
