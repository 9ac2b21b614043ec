// MX-fp8 variant of the style encoder's LSA flash attention (BASELINE config 5: "fp8 MFMA attention"; vit_set.py:52-67) on the block-scaled
// matrix instruction of gfx950, v_mfma_scale_f32_32x32x64_f8f6f4: OCP e4m3 operands, one E8M0 (power-of-two) scale per 32 elements along
// the contraction, applied by the hardware — twice the bf16 MFMA rate, a quarter of the MFMA instructions (K = 64 per instruction).
//   q (times exp(temperature) log2 e), k : scale per (token, half of the 64 head channels);
//   v                                   : scale per (head channel, 32 keys of a 64-key tile: the 32 keys ONE lane half contracts, see below);
//   P                                   : 16 p = exp2(s' + 4) in [0, 448], one scale 2^-4 for all (the + 4 rides in the accumulator's
//                                         initial value together with the softmax reference, so it costs nothing);
//   logits, row sums, the reference / redo rule and the normalisation are fp32, as in lsa_flash64_kernel (svit.hip), whose structure this
//   kernel has: 64 queries per wave (two 32-query blocks), K / V^T fragments read once for both blocks, no per-tile row maximum.
// No tensor-wide amax pass (round 2's per-tensor scales needed one over all of q, k, v before anything could be packed): the scales are local
// to a 32-element block, so ONE pack pass from the qkv GEMM's 16-bit output writes bytes + scales.
// Operand lane maps (measured with exact integer data and per-lane scales, tools/probe_mx_fp8*.hip): lane l = (r = l & 31, h = l >> 5) holds,
// in byte t of its 8 VGPRs, A[row r][k = 32 (t >> 4) + 16 h + (t & 15)] (B alike: B[k][col r]) — the two 16-byte halves of a lane belong to
// the two 32-wide scale blocks; the scale VGPR (byte 0) of lane (r, h) scales block h = k in [32 h, 32 h + 32) of row r, i.e. bytes
// 16 h .. 16 h + 15 of BOTH lanes (r, 0) and (r, 1). E8M0 127 = 1.0; C/D as the 32x32 bf16 forms.
// A precision experiment like every fp8 mode here: its deviation from the reference is measured and reported (tests/test_gpu_style.py),
// never asserted at 1e-3.
#include <float.h>

#include "conv_common.hpp"
using namespace stedm;

#define GLDS16(gptr, lptr)                                                                                  \
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr),                   \
                                   (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);   // bytes 0, 1
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);    // bytes 2, 3
  return (unsigned)w;
}

// E8M0 exponent e (byte e + 127) with amax / 2^e <= 448 (e4m3's largest finite value): frexp gives amax / 448 = m 2^ex, m in [0.5, 1)
__device__ __forceinline__ int mx_exp(float amax) {
  if (!(amax > 0.f)) return -127;
  int ex;
  (void)frexpf(amax * (1.0f / 448.0f), &ex);
  return ex < -127 ? -127 : (ex > 127 ? 127 : ex);
}

// One block = 64 tokens of one (sample, head). q / k: one thread per (token, channel half, tensor); v: one thread per (channel, key half).
// Position p = 32 s + 16 h + m of a 64-key tile of V^T holds key 32 s + 8 (m >> 2) + 4 h + (m & 3): contraction index p of the PV product is
// byte t = 16 s + m of lane half h (operand map above), and that byte of the flash kernel's P fragment is accumulator element e = m of
// sub-tile s, i.e. row (e & 3) + 8 (e >> 2) + 4 h of S^T. A scale block (32 positions) is then one 32-key sub-tile, permuted inside.
template <typename TI>
__global__ void __launch_bounds__(256) qkv_pack_mx8_kernel(const TI* __restrict__ qkv, float qscale, unsigned char* __restrict__ q8, unsigned char* __restrict__ qs,
                                                           unsigned char* __restrict__ k8, unsigned char* __restrict__ ks, unsigned char* __restrict__ v8,
                                                           unsigned char* __restrict__ vs, int Tn, int Tp, int H) {
  __shared__ float sv[64][65];          // the tile's V rows (token-major) for the transposed gather; q / k never touch LDS
  const int bh = blockIdx.x, b = bh / H, hd = bh % H;
  const int kt = blockIdx.y, t0 = kt * 64;
  const int HD = H * 64;
  typedef TI VI8 __attribute__((ext_vector_type(8)));
  {   // V tile -> LDS: one thread = 16 consecutive channels of a token (two 8-element loads)
    const int tl = threadIdx.x >> 2, c16 = (threadIdx.x & 3) * 16;
    const int t = t0 + tl;
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = 0.f;
    if (t < Tn) {
      const TI* p = qkv + ((long)b * Tn + t) * (3 * HD) + 2 * HD + hd * 64 + c16;
      const VI8 x0 = *reinterpret_cast<const VI8*>(p), x1 = *reinterpret_cast<const VI8*>(p + 8);
#pragma unroll
      for (int u = 0; u < 8; ++u) { v[u] = (float)x0[u]; v[8 + u] = (float)x1[u]; }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) sv[tl][c16 + u] = v[u];
  }
  {   // q, k rows straight from global: (token tl, channel half hf, tensor w): 32 contiguous values
    const int tl = threadIdx.x & 63, hf = (threadIdx.x >> 6) & 1, w = threadIdx.x >> 7;
    const int t = t0 + tl;
    float row[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) row[j] = 0.f;
    if (t < Tn) {
      const TI* p = qkv + ((long)b * Tn + t) * (3 * HD) + w * HD + hd * 64 + hf * 32;
      const float m = w == 0 ? qscale : 1.0f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const VI8 x = *reinterpret_cast<const VI8*>(p + 8 * g);
#pragma unroll
        for (int u = 0; u < 8; ++u) row[8 * g + u] = (float)x[u] * m;
      }
    }
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) amax = fmaxf(amax, fabsf(row[j]));
    const int e = mx_exp(amax);
    const float inv = ldexpf(1.0f, -e);
    uint4 o[2];
    unsigned* ow = reinterpret_cast<unsigned*>(o);
#pragma unroll
    for (int j = 0; j < 8; ++j) ow[j] = pack4_fp8(row[4 * j] * inv, row[4 * j + 1] * inv, row[4 * j + 2] * inv, row[4 * j + 3] * inv);
    unsigned char* d8 = (w == 0 ? q8 : k8) + ((long)bh * Tp + t0 + tl) * 64 + hf * 32;
    reinterpret_cast<uint4*>(d8)[0] = o[0];
    reinterpret_cast<uint4*>(d8)[1] = o[1];
    (w == 0 ? qs : ks)[((long)bh * Tp + t0 + tl) * 2 + hf] = (unsigned char)(e + 127);
  }
  __syncthreads();
  if (threadIdx.x < 128) {   // v: (channel d, 32-key sub-tile h = scale block)
    const int d = threadIdx.x & 63, h = threadIdx.x >> 6;
    float x[32];
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int key = 32 * h + ((j & 15) >> 2) * 8 + 4 * (j >> 4) + (j & 3);        // position 32 h + j
      x[j] = sv[key][d];
      amax = fmaxf(amax, fabsf(x[j]));
    }
    const int e = mx_exp(amax);
    const float inv = ldexpf(1.0f, -e);
    uint4 o[2];
    unsigned* ow = reinterpret_cast<unsigned*>(o);
#pragma unroll
    for (int j = 0; j < 8; ++j) ow[j] = pack4_fp8(x[4 * j] * inv, x[4 * j + 1] * inv, x[4 * j + 2] * inv, x[4 * j + 3] * inv);
    unsigned char* d8 = v8 + ((long)bh * 64 + d) * Tp + t0 + h * 32;
    reinterpret_cast<uint4*>(d8)[0] = o[0];
    reinterpret_cast<uint4*>(d8)[1] = o[1];
    vs[((long)bh * (Tp / 32) + kt * 2 + h) * 64 + d] = (unsigned char)(e + 127);
  }
}

struct Mx8Args {
  const unsigned char *q8, *qs, *k8, *ks, *v8, *vs;
  void* oh;   // [B][T][H*64], 16-bit
  int T, Tp, H;
};

template <typename T>
__global__ void __launch_bounds__(256, 2) lsa_flash64_mx8_kernel(Mx8Args a) {
  constexpr int NBUF = 3, TILE_B = 9216;          // K8 4 KiB | V8^T 4 KiB | K scales 128 B | V scales 128 B | pad
  constexpr int KS_OFF = 8192, VS_OFF = 8320;
  constexpr int O_BYTES = 4 * 32 * 65 * (int)sizeof(float);
  constexpr int LDS_B = NBUF * TILE_B > O_BYTES ? NBUF * TILE_B : O_BYTES;
  __shared__ __attribute__((aligned(1024))) unsigned char ring[LDS_B];
  float (*sO)[32][65] = reinterpret_cast<float (*)[32][65]>(ring);
  int bh, qtile;
  {
    const int nq = (a.Tp + 255) / 256, nbh = gridDim.x / nq, L = blockIdx.x;
    if ((nbh & 7) == 0) { const int x = L & 7, j = L >> 3; bh = x + 8 * (j / nq); qtile = j % nq; }
    else { bh = L % nbh; qtile = L / nbh; }
  }
  const int b = bh / a.H, hd = bh % a.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int qw = qtile * 256 + wave * 64;
  const bool live = qw < a.T;                       // wave-uniform

  v8i qf[2];
  int qsc[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    int row = qw + qb * 32 + r;
    row = row < a.Tp ? row : a.Tp - 1;
    const uint4* p = reinterpret_cast<const uint4*>(a.q8 + ((long)bh * a.Tp + row) * 64 + h * 16);      // pieces h and 2 + h
    const uint4 x0 = p[0], x1 = p[2];
    qf[qb] = v8i{(int)x0.x, (int)x0.y, (int)x0.z, (int)x0.w, (int)x1.x, (int)x1.y, (int)x1.z, (int)x1.w};
    qsc[qb] = a.qs[((long)bh * a.Tp + row) * 2 + h];
  }
  // DMA sources of this wave: chunk `wave` (16 rows) of the K8 tile and of the V8^T tile; lane l fills physical 16-B piece l & 3 of row
  // 16 wave + (l >> 2) with logical piece (l & 3) ^ ((row >> 2) & 3): conflict-free ds_read_b128 fragments on 64-B rows
  const unsigned char* const kbase = a.k8 + (long)bh * a.Tp * 64;
  const unsigned char* const vbase = a.v8 + (long)bh * 64 * a.Tp;
  const unsigned char* const ksbase = a.ks + (long)bh * a.Tp * 2;
  const unsigned char* const vsbase = a.vs + (long)bh * (a.Tp / 32) * 64;
  unsigned koff, voff;
  {
    const int row = 16 * wave + (lane >> 2);
    const int lg = (lane & 3) ^ ((row >> 2) & 3);
    koff = row * 64 + lg * 16;
    voff = row * a.Tp + lg * 16;
  }
  auto issue = [&](int kt) __attribute__((always_inline)) {
    unsigned char* dst = ring + (kt % NBUF) * TILE_B;
    GLDS16(kbase + (long)kt * 4096 + koff, dst + wave * 1024);
    GLDS16(vbase + (long)kt * 64 + voff, dst + 4096 + wave * 1024);
    if (wave == 0 && lane < 16) {
      // K scales [64 keys][2] = 128 B, then V scales [2 halves][64 channels] = 128 B: 16 lanes x 16 B, contiguous in the tile
      const unsigned char* src = lane < 8 ? ksbase + (long)kt * 128 + lane * 16 : vsbase + (long)kt * 128 + (lane - 8) * 16;
      GLDS16(src, dst + KS_OFF);
    }
  };
  f32x16 o[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[qb][d][e] = 0.f;
  // accumulator start of S^T: 4 - m_ref (16 p is what the bytes carry; m_ref: the softmax reference of lsa_flash64_kernel)
  constexpr float kPShift = 4.0f;
  constexpr int kPScale = 127 - 4;
  f32x16 cinit[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int e = 0; e < 16; ++e) cinit[qb][e] = kPShift;
  float l_run[2] = {0.f, 0.f}, m_ref[2] = {0.f, 0.f};
  const int ntiles = (a.T + 63) / 64;
  // fragment byte offsets in a tile: K row = 32 sub + r, V^T row = 32 dblk + r; logical 16-B pieces h and 2 + h (operand map above)
  unsigned kb[2][2], vb[2][2], ksb[2], vsb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = i * 32 + r;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      kb[i][j] = row * 64 + (((h + 2 * j) ^ ((row >> 2) & 3)) << 4);
      vb[i][j] = 4096 + row * 64 + (((h + 2 * j) ^ ((row >> 2) & 3)) << 4);
    }
    ksb[i] = KS_OFF + row * 2 + h;
    vsb[i] = VS_OFF + h * 64 + row;
  }
  auto masks = [&](f32x16 (&s)[2], const int qb, const int kt) __attribute__((always_inline)) {
    const int q0 = qw + qb * 32, qidx = q0 + r;
    if ((q0 >> 6) == kt) {
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
  };
  auto move_ref = [&](f32x16 (&s)[2], const int qb, const bool first) __attribute__((always_inline)) {
    float mx = s[0][0];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[sub][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    mx -= kPShift;                                 // (the logits arrive with the + 4 of the byte scale)
    const float delta = first ? fmaxf(mx, -30000.f) : fmaxf(mx, 0.f);
    const float alpha = __builtin_amdgcn_exp2f(-delta);
    m_ref[qb] += delta;
#pragma unroll
    for (int e = 0; e < 16; ++e) cinit[qb][e] = kPShift - m_ref[qb];
    l_run[qb] *= alpha;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[qb][d][e] *= alpha;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[sub][e] -= delta;
  };
  // p' = 16 p = exp2(s'), row sum, bytes: element t = 16 sub + e of a lane is key 32 sub + 8 (e >> 2) + 4 h + (e & 3) = position
  // 32 sub + 16 h + e of the tile's V^T image = the contraction index of byte t of the lane's B fragment
  auto softmax = [&](f32x16 (&s)[2], v8i& p8, float& rs) __attribute__((always_inline)) {
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      float pv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { pv[u] = __builtin_amdgcn_exp2f(s[n >> 2][(n & 3) * 4 + u]); rs += pv[u]; }
      p8[n] = (int)pack4_fp8(pv[0], pv[1], pv[2], pv[3]);
    }
  };
  // no per-tile row maximum: the row sum of 16 p bounds every byte (e4m3 carries 448); beyond kLimit the tile is redone with the reference moved
  constexpr float kLimit = 448.0f;     // (a sum of 16 p that fits e4m3 bounds each of them; a false alarm redoes the tile unchanged)
  v8i kf[2], vf[2];
  int ksc[2], vsc[2];
  auto qk = [&](f32x16 (&s)[2], const int qb) __attribute__((always_inline)) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
      s[sub] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf[sub], qf[qb], cinit[qb], 0, 0, 0, ksc[sub], 0, qsc[qb]);
  };
  auto load8 = [&](const unsigned char* tb, const unsigned (&off)[2]) __attribute__((always_inline)) {
    const uint4 x0 = *reinterpret_cast<const uint4*>(tb + off[0]);
    const uint4 x1 = *reinterpret_cast<const uint4*>(tb + off[1]);
    return v8i{(int)x0.x, (int)x0.y, (int)x0.z, (int)x0.w, (int)x1.x, (int)x1.y, (int)x1.z, (int)x1.w};
  };

  issue(0);
  if (ntiles > 1) issue(1);
  for (int kt = 0; kt < ntiles; ++kt) {
    // this wave's share of tile kt has landed (its newest DMAs, 2 or — wave 0 — 3, belong to tile kt + 1 when there is one)
    if (kt + 1 < ntiles) { if (wave == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // tile kt is complete in LDS; every wave is done with tile kt - 1, whose buffer takes the next tile fetched
    if (kt + NBUF - 1 < ntiles) issue(kt + NBUF - 1);
    if (!live) continue;            // a wave without queries (tail of the last 256-query tile) only feeds the ring
    const unsigned char* tb = ring + (kt % NBUF) * TILE_B;
#pragma unroll
    for (int i = 0; i < 2; ++i) { kf[i] = load8(tb, kb[i]); ksc[i] = tb[ksb[i]]; }
    f32x16 s0[2], s1[2];
    qk(s0, 0);
    qk(s1, 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) { vf[i] = load8(tb, vb[i]); vsc[i] = tb[vsb[i]]; }
    v8i p0, p1;
    float rs0 = 0.f, rs1 = 0.f;
    // ---- block 0 (block 1's S^T MFMAs run beneath its exponentials)
    masks(s0, 0, kt);
    if (kt == 0) move_ref(s0, 0, true);
    softmax(s0, p0, rs0);
    if (__builtin_amdgcn_ballot_w64(!(rs0 <= kLimit))) {
      qk(s0, 0); masks(s0, 0, kt); move_ref(s0, 0, false); rs0 = 0.f; softmax(s0, p0, rs0);
    }
    l_run[0] += rs0;
#pragma unroll
    for (int d = 0; d < 2; ++d) o[0][d] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf[d], p0, o[0][d], 0, 0, 0, vsc[d], 0, kPScale);
    // ---- block 1 (block 0's O^T MFMAs run beneath its exponentials)
    masks(s1, 1, kt);
    if (kt == 0) move_ref(s1, 1, true);
    softmax(s1, p1, rs1);
    if (__builtin_amdgcn_ballot_w64(!(rs1 <= kLimit))) {
      qk(s1, 1); masks(s1, 1, kt); move_ref(s1, 1, false); rs1 = 0.f; softmax(s1, p1, rs1);
    }
    l_run[1] += rs1;
#pragma unroll
    for (int d = 0; d < 2; ++d) o[1][d] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf[d], p1, o[1][d], 0, 0, 0, vsc[d], 0, kPScale);
  }
  // ---- epilogue: O^T / l through LDS, one query block after the other ([B][T][H*64]); l counted 16 p, O the true p
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float lr = l_run[qb];
    lr += __shfl_xor(lr, 32, 64);
    const float inv = live ? 16.0f / lr : 0.f;
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) sO[wave][r][d * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] = o[qb][d][e] * inv;
    __syncthreads();
    const int row = lane >> 1, half = lane & 1;
    const int t = qw + qb * 32 + row;
    if (t < a.T) {
      T* oh = reinterpret_cast<T*>(a.oh) + ((long)b * a.T + t) * (a.H * 64) + hd * 64 + half * 32;
#pragma unroll
      for (int j = 0; j < 32; ++j) oh[j] = (T)sO[wave][row][half * 32 + j];
    }
  }
}

}  // namespace

extern "C" int stedm_qkv_pack_mx8(const void* qkv, int qkv_is16, float qscale, void* q8, void* qs, void* k8, void* ks, void* vt8, void* vs, int B, int T,
                                  int Tp, int heads, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(qkv && q8 && qs && k8 && ks && vt8 && vs, "qkv_pack_mx8: null pointer");
  STEDM_CHECK_ARG(B > 0 && heads > 0 && T > 1 && Tp >= T && Tp % 128 == 0, "qkv_pack_mx8: need Tp %% 128 == 0, Tp >= T > 1");
  dim3 grid(B * heads, Tp / 64);
  hipStream_t st = as_stream(stream);
  auto* p8 = reinterpret_cast<unsigned char*>(q8);
  auto *pqs = reinterpret_cast<unsigned char*>(qs), *pk8 = reinterpret_cast<unsigned char*>(k8), *pks = reinterpret_cast<unsigned char*>(ks);
  auto *pv8 = reinterpret_cast<unsigned char*>(vt8), *pvs = reinterpret_cast<unsigned char*>(vs);
  if (!qkv_is16) qkv_pack_mx8_kernel<float><<<grid, 256, 0, st>>>(reinterpret_cast<const float*>(qkv), qscale, p8, pqs, pk8, pks, pv8, pvs, T, Tp, heads);
  else if (mm_dtype == STEDM_F16) qkv_pack_mx8_kernel<_Float16><<<grid, 256, 0, st>>>(reinterpret_cast<const _Float16*>(qkv), qscale, p8, pqs, pk8, pks, pv8, pvs, T, Tp, heads);
  else qkv_pack_mx8_kernel<__bf16><<<grid, 256, 0, st>>>(reinterpret_cast<const __bf16*>(qkv), qscale, p8, pqs, pk8, pks, pv8, pvs, T, Tp, heads);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_lsa_flash_mx8(const void* q8, const void* qs, const void* k8, const void* ks, const void* vt8, const void* vs, void* out16, int B,
                                   int T, int Tp, int heads, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(q8 && qs && k8 && ks && vt8 && vs && out16, "lsa_flash_mx8: null pointer");
  STEDM_CHECK_ARG(Tp % 128 == 0 && Tp >= T && T > 1, "lsa_flash_mx8: need Tp %% 128 == 0, Tp >= T > 1");
  Mx8Args a{reinterpret_cast<const unsigned char*>(q8), reinterpret_cast<const unsigned char*>(qs), reinterpret_cast<const unsigned char*>(k8),
            reinterpret_cast<const unsigned char*>(ks), reinterpret_cast<const unsigned char*>(vt8), reinterpret_cast<const unsigned char*>(vs),
            out16, T, Tp, heads};
  const dim3 grid(B * heads * ((Tp + 255) / 256));
  if (mm_dtype == STEDM_F16) lsa_flash64_mx8_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>(a);
  else lsa_flash64_mx8_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>(a);
  STEDM_LAUNCH_CHECK();
  return 0;
}
