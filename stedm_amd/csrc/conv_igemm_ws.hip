// v2 fused implicit-GEMM convolution: warp-specialised, LDS-DMA weight tiles, double-buffered halo patch.
//
// Same math and HBM layout as v1 (conv_igemm.hip). What changes is who does what inside a workgroup:
//
//   512 threads = 8 waves. Waves 0-3 (one per SIMD) are COMPUTE waves: they only read MFMA fragments
//   from LDS and issue MFMAs; each owns a (WM*32) x 64 slab of the (2*WM*32) x 128 output tile.
//   Waves 4-7 (the SIMD partners of 0-3) are LOADER waves: while tap t of chunk c is being multiplied
//   they (a) DMA the weight tile of step t+1 straight into LDS with global_load_lds_dwordx4 (no VGPR
//   staging; the 16-B pieces of a row are XOR-swizzled through the *source* address so fragment reads
//   are bank-conflict free on an unpadded image), and (b) fetch 1/9 of the NEXT chunk's haloed input
//   patch, apply the GroupNorm affine + SiLU, convert to the 16-bit operand format (hi/lo planes for
//   the split-precision mode) and store it into the other patch buffer. VALU/transcendental work of the
//   loaders co-issues with the partner wave's MFMAs; one workgroup barrier per tap is the only sync.
//
//   Tile 256 x 128 (WM = 4) halves the weight bytes per FLOP versus v1's 128 x 128; WM = 2 keeps
//   128 x 128 for problems with few output pixels so that >= 1 tile per CU exists.
#include "conv_common.hpp"
using namespace stedm;

#define GLDS16(gptr, lptr)                                                                                  \
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr),                   \
                                   (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

template <int BKC, int NPASS, typename T, int WM, int NBUF>
__global__ void __launch_bounds__(512, 2) conv_ws_kernel(const ConvParams p) {
  using V8 = typename MM<T>::V8;
  using V4 = typename MM<T>::V4;
  constexpr int BMT = 2 * WM * 32;           // M tile
  constexpr int AST = BKC * 2 + 16;          // bytes per patch position (padded: conflict-free b128 reads)
  constexpr int BROW = BKC * 2;              // bytes per weight row (unpadded, swizzled)
  constexpr int NPL = NPASS == 3 ? 2 : 1;
  constexpr int QC = BKC / 4;                // float4 quads per patch position
  constexpr int POSL = 256 / QC;             // patch positions per loader sweep
  constexpr int PCS = BKC / 8;               // 16-B pieces per weight row
  constexpr int RPB = 16 / PCS;              // weight rows per 256-B LDS bank row
  constexpr int NBI = BN * PCS / 64;         // 1-KiB DMA instructions per weight plane
  constexpr int NBG = NBI / 4 * NPL;         // DMA instructions per loader wave per step
  constexpr int KSTEPS = BKC / 16;
  constexpr int b_plane = BN * BROW;
  constexpr int SPS_MAX = 4;                 // patch sweeps a loader keeps in flight across a barrier

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int a_plane = p.NP * AST;
  const int a_buf = NPL * a_plane;
  unsigned char* sA = smem;                  // [2][NPL][NP][AST]
  unsigned char* sB = smem + 2 * a_buf;      // [NBUF][NPL][BN][BROW]
  int* sIdx = reinterpret_cast<int*>(sB + NBUF * NPL * b_plane);   // [taps][BMT] byte offsets of the patch rows
  int* sPix = sIdx + p.taps * BMT;           // [NP] source pixel index for src1 (-1: padding)
  int* sPix2 = sPix + p.NP;                  // [NP] same for src2 (batch modulo applied)
  int* sSmp = sPix2 + p.NP;                  // [NP] sample index b (scale/shift row)

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

  // ---- tables (all 512 threads, once): every div/mod of the tile geometry happens here and nowhere else
  for (int i = tid; i < taps * BMT; i += 512) {       // gather table: patch row read by output pixel ml at tap
    const int tap = i / BMT, ml = i - tap * BMT;
    int idx;
    if (is1x1) idx = ml;
    else {
      int s_, yl, x, y;
      if (p.whole) { s_ = ml / p.HWout; const int rem = ml - s_ * p.HWout; yl = rem / p.Wout; x = rem - yl * p.Wout; y = yl; }
      else { s_ = 0; yl = ml / p.Wout; x = ml - yl * p.Wout; y = yo0 + yl; }
      const int dy = tap / 3, dx = tap - dy * 3;
      int prow, pcol;
      if (a.mode == STEDM_CONV_S1) { prow = yl + dy; pcol = x + dx; }
      else if (a.mode == STEDM_CONV_DOWN) { prow = 2 * yl + dy; pcol = 2 * x + dx; }
      else { prow = ((y + dy - 1) >> 1) - srow0; pcol = ((x + dx - 1) >> 1) + 1; }
      idx = (s_ * p.PRs + prow) * p.PW + pcol;
    }
    sIdx[i] = idx * AST;
  }
  for (int pos = tid; pos < p.NP; pos += 512) {        // patch position -> source pixel / sample
    int b, pix = -1, pix2 = -1;
    if (is1x1) {
      const int m = m0 + pos;
      b = m < p.M ? m / p.HWout : 0;
      if (m < p.M) {
        pix = m;
        const int bs = a.src2_bmod > 0 ? b % a.src2_bmod : b;
        pix2 = bs * p.HWout + (m - b * p.HWout);
      }
    } else {
      const int s_ = pos / (p.PRs * p.PW), rem = pos - s_ * (p.PRs * p.PW);
      const int prow = rem / p.PW, pcol = rem - prow * p.PW;
      b = b0 + s_;
      const int sy = srow0 + prow, sx = pcol - 1;
      if (b < a.B && sy >= 0 && sy < a.Hin && sx >= 0 && sx < a.Win) {
        pix = (b * a.Hin + sy) * a.Win + sx;
        const int bs = a.src2_bmod > 0 ? b % a.src2_bmod : b;
        pix2 = (bs * a.Hin + sy) * a.Win + sx;
      }
    }
    sPix[pos] = pix; sPix2[pos] = pix2; sSmp[pos] = b;
  }
  __syncthreads();

  if (is_loader) {
    // =============================================================================== LOADER WAVES
    const int lt = tid - 256;
    const int lw = wave - 4;
    const int q4 = (lt % QC) * 4, pl0 = lt / QC;
    const int nsweeps = (p.NP + POSL - 1) / POSL;
    const int sps = (nsweeps + taps - 1) / taps;   // sweeps per tap-slice
    const bool piped = sps <= SPS_MAX;             // slice loads stay in flight across the step barrier
    const bool affine = a.scale != nullptr;
    const bool act = a.act == 1;

    // weight DMA: per-lane element offset of (row, swizzled piece) for each of this wave's DMA instructions
    long wrow[NBI / 4];
#pragma unroll
    for (int jj = 0; jj < NBI / 4; ++jj) {
      const int piece = (lw + jj * 4) * 64 + lane;
      const int row = piece / PCS, phys = piece % PCS;
      const int logical = phys ^ ((row / RPB) % PCS);
      int n = n0 + row;
      n = n < a.cout ? n : a.cout - 1;             // masked columns read a valid row; discarded in the epilogue
      wrow[jj] = (long)n * taps * p.Cin + logical * 8;
    }
    auto issue_b = [&](int step) {
      const int chunk = step / taps, tap = step - chunk * taps;
      const long so = (long)tap * p.Cin + chunk * BKC;
      unsigned char* dstb = sB + ((step % NBUF) * NPL) * b_plane;
#pragma unroll
      for (int jj = 0; jj < NBI / 4; ++jj) {
        const int j = lw + jj * 4;                  // wave-uniform
        GLDS16(reinterpret_cast<const uint16_t*>(a.w_hi) + wrow[jj] + so, dstb + j * 1024);
        if (NPL == 2) GLDS16(reinterpret_cast<const uint16_t*>(a.w_lo) + wrow[jj] + so, dstb + b_plane + j * 1024);
      }
    };

    // patch slice state: up to SPS_MAX 16-B loads in flight, with the position / sample they belong to
    float4 v[SPS_MAX];
    int vpos[SPS_MAX], vsmp[SPS_MAX];   // vpos < 0: no store; vsmp < 0: padding (store zeros)
    int fc0 = 0;                        // first channel of the fetched chunk
    int cur_b = -1;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);

    // fetch sweeps [k0, k0+n) of `chunk`: exactly one global load per sweep slot and lane (padding reads the tensor base)
    auto fetch = [&](int chunk, int k0, int n) {
      fc0 = chunk * BKC;
      const bool second = fc0 >= a.c1;
      const float* src = second ? a.src2 : a.src1;
      const int Cs = second ? a.c2 : a.c1;
      const int cs = (second ? fc0 - a.c1 : fc0) + q4;
      const int* tab = second ? sPix2 : sPix;
#pragma unroll
      for (int u = 0; u < SPS_MAX; ++u) {
        vpos[u] = -1; vsmp[u] = -1;
        if (u < n) {
          const int pos = pl0 + (k0 + u) * POSL;
          long goff = cs;
          if (pos < p.NP) {
            vpos[u] = pos;
            const int pix = tab[pos];
            if (pix >= 0) { vsmp[u] = sSmp[pos]; goff = (long)pix * Cs + cs; }
          }
          v[u] = *reinterpret_cast<const float4*>(src + goff);
        }
      }
    };
    auto commit = [&](unsigned char* dstA) {
#pragma unroll
      for (int u = 0; u < SPS_MAX; ++u) {
        if (vpos[u] < 0) continue;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vsmp[u] >= 0) {
          x = v[u];
          if (affine) {
            if (vsmp[u] != cur_b) {
              cur_b = vsmp[u];
              sc = *reinterpret_cast<const float4*>(a.scale + (long)cur_b * p.Cin + fc0 + q4);
              sh = *reinterpret_cast<const float4*>(a.shift + (long)cur_b * p.Cin + fc0 + q4);
            }
            x.x = fmaf(x.x, sc.x, sh.x); x.y = fmaf(x.y, sc.y, sh.y);
            x.z = fmaf(x.z, sc.z, sh.z); x.w = fmaf(x.w, sc.w, sh.w);
          }
          if (act) { x.x = silu_f(x.x); x.y = silu_f(x.y); x.z = silu_f(x.z); x.w = silu_f(x.w); }
        }
        V4 hi;
        hi[0] = (T)x.x; hi[1] = (T)x.y; hi[2] = (T)x.z; hi[3] = (T)x.w;
        *reinterpret_cast<V4*>(dstA + vpos[u] * AST + q4 * 2) = hi;
        if (NPL == 2) {
          V4 lo;
          lo[0] = (T)(x.x - (float)hi[0]); lo[1] = (T)(x.y - (float)hi[1]);
          lo[2] = (T)(x.z - (float)hi[2]); lo[3] = (T)(x.w - (float)hi[3]);
          *reinterpret_cast<V4*>(dstA + a_plane + vpos[u] * AST + q4 * 2) = lo;
        }
      }
    };
    auto stage_sync = [&](int chunk, int k0, int n, unsigned char* dstA) {   // fetch + commit, nothing left in flight
      for (int k = 0; k < n; k += SPS_MAX) {
        fetch(chunk, k0 + k, n - k < SPS_MAX ? n - k : SPS_MAX);
        commit(dstA);
      }
    };
#define WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
    auto wait_keep = [&](int keep) {   // all but the `keep` youngest vector-memory ops of this wave have completed
      switch (keep) {
        case 0: WAIT_VM(0); break; case 1: WAIT_VM(1); break; case 2: WAIT_VM(2); break; case 3: WAIT_VM(3); break;
        case 4: WAIT_VM(4); break; case 5: WAIT_VM(5); break; case 6: WAIT_VM(6); break; case 7: WAIT_VM(7); break;
        case 8: WAIT_VM(8); break; case 9: WAIT_VM(9); break; case 10: WAIT_VM(10); break; case 11: WAIT_VM(11); break;
        default: WAIT_VM(12); break;
      }
    };

    // ---- prologue: whole patch of chunk 0, weight tile(s) of step 0 (and 1), first slice of chunk 1 in flight
    issue_b(0);
    if (NBUF == 3 && nsteps > 1) issue_b(1);
    cur_b = -1;
    stage_sync(0, 0, nsweeps, sA);
    bool pending = false;            // a fetched-but-uncommitted slice is held in v[]
    if (piped && nchunks > 1 && !(p.dbg & 2)) { cur_b = -1; fetch(1, 0, sps); pending = true; }
    wait_keep(pending ? sps : 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();

    for (int step = 0; step < nsteps; ++step) {
      const int chunk = step / taps, tap = step - chunk * taps;
      unsigned char* dst = sA + ((chunk + 1) & 1) * a_buf;
      // (1) commit the slice fetched during the previous step (its loads are the oldest outstanding ops)
      if (pending) { commit(dst); pending = false; }
      // (2) DMA of a later weight tile: NBUF-1 steps ahead
      const int bstep = step + NBUF - 1;
      const bool issued_b = bstep < nsteps && !(p.dbg & 1);
      if (issued_b) issue_b(bstep);
      // (3) register loads of the next patch slice (slice tap+1 of chunk+1, or slice 0 of chunk+2)
      int nfly = 0;
      if (!(p.dbg & 2)) {
        if (piped) {
          if (tap + 1 < taps) { if (chunk + 1 < nchunks) { fetch(chunk + 1, (tap + 1) * sps, sps); pending = true; nfly = sps; } }
          else if (chunk + 2 < nchunks) { cur_b = -1; fetch(chunk + 2, 0, sps); pending = true; nfly = sps; }
        } else if (chunk + 1 < nchunks) {
          if (tap == 0) cur_b = -1;
          stage_sync(chunk + 1, tap * sps, sps, dst);
        }
      }
      // (4) everything older than what this step issued must have landed: with NBUF == 3 that is the weight tile of
      //     step+1 (issued one step ago); with NBUF == 2 the tile issued in (2) must itself land -> do not count it
      wait_keep(nfly + ((NBUF == 3 && issued_b) ? NBG : 0));
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): patch stores of this step are in LDS
      __builtin_amdgcn_s_barrier();
    }
#undef WAIT_VM
    return;
  }

  // ================================================================================= COMPUTE WAVES
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  // weight fragment rows and their swizzle keys
  int brow_off[2], bswz[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wn * 64 + j * 32 + r;
    brow_off[j] = row * BROW;
    bswz[j] = (row / RPB) % PCS;
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
  __builtin_amdgcn_s_barrier();   // prologue data landed
  for (int step = 0; step < nsteps; ++step) {
    const int chunk = step / taps, tap = step - chunk * taps;
    // LDS byte addresses (low 32 bits of a shared pointer are the LDS offset)
    const unsigned Abase = (unsigned)(uintptr_t)(sA + (chunk & 1) * a_buf) + h * 16;
    const unsigned Bbase = (unsigned)(uintptr_t)(sB + ((step % NBUF) * NPL) * b_plane);
    unsigned pa[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) pa[i] = Abase + myIdx[tap * BMT + i * 32];

    V8 ah[2][WM], bh[2][2], al[2][WM], bl[2][2];
    // Fragment reads are inline asm so that hipcc cannot sink them next to their consumers: the reads of k-step
    // ks+1 are issued BEFORE the MFMAs of k-step ks and retired by a counted lgkmcnt (LDS returns in order).
#define LDSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LOAD_FRAGS(KS, ST)                                                                             \
    {                                                                                                  \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                  \
        const unsigned ba = Bbase + brow_off[j] + ((((KS) * 2 + h) ^ bswz[j]) * 16);                   \
        LDSR(bh[ST][j], ba, 0);                                                                        \
        if (NPASS == 3) LDSR(bl[ST][j], ba + b_plane, 0);                                              \
      }                                                                                                \
      _Pragma("unroll") for (int i = 0; i < WM; ++i) {                                                 \
        LDSR(ah[ST][i], pa[i], (KS) * 32);                                                             \
        if (NPASS == 3) LDSR(al[ST][i], pa[i] + a_plane, (KS) * 32);                                   \
      }                                                                                                \
    }
    constexpr int NRD = (WM + 2) * NPL;   // reads per k-step
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

  // ---- epilogue: bias + emb broadcast + residual, NHWC store
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + r;
    if (n >= a.cout) continue;
    const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * (WM * 32) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v = acc[i][j][e] + bv;
        if (a.emb) v += a.emb[(long)(m / p.HWout) * a.emb_bstride + n];
        const long o = (long)m * a.cout + n;
        if (a.res) v += a.res[o];
        a.out[o] = v;
      }
    }
  }
}

template <int BKC, int NPASS, typename T, int WM, int NBUF>
static size_t ws_lds_bytes(const ConvParams& p) {
  constexpr int AST = BKC * 2 + 16, NPL = NPASS == 3 ? 2 : 1;
  return (size_t)2 * NPL * p.NP * AST + (size_t)NBUF * NPL * BN * BKC * 2 + ((size_t)p.taps * (2 * WM * 32) + 3 * (size_t)p.NP) * sizeof(int);
}

template <int BKC, int NPASS, typename T, int WM, int NBUF>
static int ws_launch(const ConvParams& p, hipStream_t st) {
  const size_t lds = ws_lds_bytes<BKC, NPASS, T, WM, NBUF>(p);
  auto k = conv_ws_kernel<BKC, NPASS, T, WM, NBUF>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("conv_igemm(ws): hipFuncSetAttribute(%zu) failed: %s", lds, hipGetErrorString(e));
      return 2;
    }
  }
  k<<<p.tiles_m * p.tiles_n, 512, lds, st>>>(p);
  STEDM_LAUNCH_CHECK();
  return 0;
}

template <int BKC, int NPASS, typename T, int WM>
static int ws_try(ConvParams& p, const ConvParams& q, hipStream_t st) {
  constexpr size_t LDS_MAX = 160 * 1024;
  if (ws_lds_bytes<BKC, NPASS, T, WM, 3>(q) <= LDS_MAX) { p = q; return ws_launch<BKC, NPASS, T, WM, 3>(p, st); }
  if (ws_lds_bytes<BKC, NPASS, T, WM, 2>(q) <= LDS_MAX) { p = q; return ws_launch<BKC, NPASS, T, WM, 2>(p, st); }
  return -1;
}

template <int NPASS, typename T>
static int ws_pick(ConvParams& p, hipStream_t st) {
  const stedm_conv_args& a = p.a;
  static int cus = 0;
  if (cus == 0) { cus = stedm_device_cus(); if (cus <= 0) cus = 256; }
  const bool c64 = (NPASS == 1) && p.Cin % 64 == 0 && (a.c2 == 0 || a.c1 % 64 == 0);
  // prefer the 256-row tile when it still yields at least ~0.75 tiles per CU, else the 128-row tile
  for (int wm4 = 1; wm4 >= 0; --wm4) {
    const int bm = wm4 ? 256 : 128;
    ConvParams q = p;
    if (!conv_geometry(q, bm)) continue;
    if (wm4 && (long)q.tiles_m * q.tiles_n * 4 < (long)cus * 3) continue;
    int rc = -1;
    if (NPASS == 1 && c64) rc = wm4 ? ws_try<64, 1, T, 4>(p, q, st) : ws_try<64, 1, T, 2>(p, q, st);
    if (rc >= 0) return rc;
    rc = wm4 ? ws_try<32, NPASS, T, 4>(p, q, st) : ws_try<32, NPASS, T, 2>(p, q, st);
    if (rc >= 0) return rc;
  }
  return -1;
}

int stedm::conv_launch_ws(ConvParams& p, hipStream_t st) {
  const stedm_conv_args& a = p.a;
  if (a.mode == STEDM_CONV_DOWN) return -1;   // large stride-2 patches: v1
  const bool f16 = a.mm_dtype == STEDM_F16;
  if (a.npass == 3) return f16 ? ws_pick<3, _Float16>(p, st) : ws_pick<3, __bf16>(p, st);
  return f16 ? ws_pick<1, _Float16>(p, st) : ws_pick<1, __bf16>(p, st);
}
