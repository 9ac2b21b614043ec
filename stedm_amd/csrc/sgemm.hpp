// Tiled fp32 GEMM shared by stedm_linear (misc.hip) and stedm_gemm_f32 (bwd.hip): the embedding path of the U-Net (time_embed, the ResBlocks'
// emb_layers as one stacked Linear, openaimodel.py:231-237, 529-534) and its backward, Agg_Linear (agg_blocks.py:14-18). fp32 throughout (these
// Linears carry the timestep / style conditioning; the reference runs them in fp32 outside autocast), products and sums on the vector pipe.
//   C[M][N] = epi( alpha * op(A) op(B) ),  op(A)[m][k] = ta ? A[k lda + m] : act_in(A[m lda + k]),  op(B)[k][n] = tb ? B[n ldb + k] : B[k ldb + n]
// Block tile TM x TN by 256 threads (16 x 16, each (TM/16) x (TN/16) outputs read from LDS as whole 16-B rows), K step 32 with the next step's
// operands prefetched into registers across the products (a block with few K steps is one dependent global-load chain; the round-4 kernels
// walked K 16 / 32 at a time without prefetch and measured 40 - 100 us on problems of 0.03 - 0.7 GFLOP). Split K over blockIdx.z into `part`
// (fixed-order reduce by the caller) as before.
#pragma once
#include "common.hpp"

namespace stedm {

struct SgemmArgs {
  const float* A; long lda; int ta;
  const float* B; long ldb; int tb;
  float* C; long ldc;
  int M, N, K;
  float alpha, beta;
  int kchunk;            // K range of a blockIdx.z slice (gridDim.z > 1: partial sums go to part[z][M][N])
  float* part;
  const float* bias;     // [N] or NULL, added before act_out (gridDim.z == 1 only)
  int act_in, act_out;   // 0 none, 1 SiLU, 2 ReLU (act_in on A's elements)
};

__device__ __forceinline__ float sgemm_act(float v, int act) {
  return act == 1 ? silu_f(v) : (act == 2 ? fmaxf(v, 0.f) : v);
}

template <int TM, int TN, int KS>
__global__ void __launch_bounds__(256) sgemm_kernel(const SgemmArgs a) {
  constexpr int RM = TM / 16, RN = TN / 16, PA = TM * KS / 256, PB = TN * KS / 256;
  static_assert((TM + TN + 8) * KS * 4 <= 64 * 1024, "static LDS");
  static_assert(RM == 1 || RM == 4, "row register tile is 1 or 4");
  static_assert(RN == 1 || RN == 4, "column register tile is 1 or 4");
  __shared__ __attribute__((aligned(16))) float sa[KS][TM + 4];
  __shared__ __attribute__((aligned(16))) float sb[KS][TN + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int kbeg = blockIdx.z * a.kchunk;
  const int kend = gridDim.z > 1 ? min(a.K, kbeg + a.kchunk) : a.K;
  float acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) acc[i][j] = 0.f;
  float pa[PA], pb[PB];
  // element e of a tile -> (k, m): the index that is contiguous in memory runs fastest over the threads
  // (loads are unconditional from clamped addresses and the zero fill / activation happen when the registers go to LDS: a load inside a
  // bounds branch makes the compiler wait for it at the join, which turned a step's 40 independent loads into 40 round trips)
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int e = tid + i * 256;
      const int kk = a.ta ? e / TM : e % KS, m = a.ta ? e % TM : e / KS;
      const int mc = min(m0 + m, a.M - 1), kc = min(k0 + kk, kend - 1);
      pa[i] = a.ta ? a.A[(long)kc * a.lda + mc] : a.A[(long)mc * a.lda + kc];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int e = tid + i * 256;
      const int kk = a.tb ? e % KS : e / TN, n = a.tb ? e / KS : e % TN;
      const int nc = min(n0 + n, a.N - 1), kc = min(k0 + kk, kend - 1);
      pb[i] = a.tb ? a.B[(long)nc * a.ldb + kc] : a.B[(long)kc * a.ldb + nc];
    }
  };
  auto stash = [&](int k0) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int e = tid + i * 256;
      const int kk = a.ta ? e / TM : e % KS, m = a.ta ? e % TM : e / KS;
      const float v = (m0 + m < a.M && k0 + kk < kend) ? pa[i] : 0.f;
      sa[kk][m] = a.act_in ? sgemm_act(v, a.act_in) : v;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int e = tid + i * 256;
      const int kk = a.tb ? e % KS : e / TN, n = a.tb ? e / KS : e % TN;
      sb[kk][n] = (n0 + n < a.N && k0 + kk < kend) ? pb[i] : 0.f;
    }
  };
  if (kbeg < kend) fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += KS) {
    __syncthreads();            // the previous step's products have read the tiles
    stash(k0);
    __syncthreads();
    if (k0 + KS < kend) fetch(k0 + KS);
#pragma unroll 16
    for (int kk = 0; kk < KS; ++kk) {
      float av[RM], bv[RN];
      if constexpr (RM == 4) {
        const float4 t = *reinterpret_cast<const float4*>(&sa[kk][ty * 4]);
        av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w;
      } else av[0] = sa[kk][ty];
      if constexpr (RN == 4) {
        const float4 t = *reinterpret_cast<const float4*>(&sb[kk][tx * 4]);
        bv[0] = t.x; bv[1] = t.y; bv[2] = t.z; bv[3] = t.w;
      } else bv[0] = sb[kk][tx];
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) {
      const int m = m0 + ty * RM + i, n = n0 + tx * RN + j;
      if (m < a.M && n < a.N) {
        if (gridDim.z > 1) a.part[((long)blockIdx.z * a.M + m) * a.N + n] = acc[i][j];
        else {
          float v = a.alpha * acc[i][j] + (a.beta != 0.f ? a.beta * a.C[(long)m * a.ldc + n] : 0.f);
          if (a.bias) v += a.bias[n];
          a.C[(long)m * a.ldc + n] = a.act_out ? sgemm_act(v, a.act_out) : v;
        }
      }
    }
}

// Tile choice. A block is one chain of K / KS dependent (prefetched) global loads of ~2.5 us each, so a problem with few tiles wants long K
// steps and narrow tiles (more blocks, each re-reading the small operand from L2), a problem with many tiles the 64 x 64 tile's 16 products per
// two LDS reads. `ks` = split-K slices (gridDim.z).
static inline void sgemm_launch(const SgemmArgs& a, int ks, hipStream_t st) {
  const long t64 = (long)((a.M + 63) / 64) * ((a.N + 63) / 64) * ks;
  if (a.M <= 16 && a.N <= 16) {
    sgemm_kernel<16, 16, 128><<<dim3((a.N + 15) / 16, (a.M + 15) / 16, ks), 256, 0, st>>>(a);
  } else if (t64 >= 512) {
    sgemm_kernel<64, 64, 32><<<dim3((a.N + 63) / 64, (a.M + 63) / 64, ks), 256, 0, st>>>(a);
  } else if (t64 >= 128) {
    sgemm_kernel<64, 64, 64><<<dim3((a.N + 63) / 64, (a.M + 63) / 64, ks), 256, 0, st>>>(a);
  } else if (a.M <= 16 || (a.N >= 64 && a.N > a.M && a.M < 64)) {
    sgemm_kernel<16, 64, 128><<<dim3((a.N + 63) / 64, (a.M + 15) / 16, ks), 256, 0, st>>>(a);
  } else if (a.N >= a.M || a.N <= 16) {
    sgemm_kernel<64, 16, 128><<<dim3((a.N + 15) / 16, (a.M + 63) / 64, ks), 256, 0, st>>>(a);
  } else {
    sgemm_kernel<16, 64, 128><<<dim3((a.N + 63) / 64, (a.M + 15) / 16, ks), 256, 0, st>>>(a);
  }
}

}  // namespace stedm
