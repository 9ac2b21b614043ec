// Shared helpers for libstedm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/stedm_hip.h"

namespace stedm {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

#define STEDM_CHECK_ARG(cond, ...)      \
  do {                                  \
    if (!(cond)) {                      \
      ::stedm::set_error(__VA_ARGS__);  \
      return 1;                         \
    }                                   \
  } while (0)

#define STEDM_HIP_TRY(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ::stedm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return 2;                                                                          \
    }                                                                                    \
  } while (0)

// after a kernel launch
#define STEDM_LAUNCH_CHECK() STEDM_HIP_TRY(hipGetLastError())

// fp16 operand range guard (stedm_f16_guard_set, include/stedm_hip.h): device address of the flag word of the current device, or nullptr.
unsigned* f16_guard_flag();

// nonzero iff one of the two / four 16-bit values packed in `w` / `q` is an fp16 inf or NaN (exponent 11111: adding one at the exponent's
// lowest bit carries into the cleared sign position). For T = __bf16 the answer is a compile-time 0 and the callers' code folds away.
template <typename T>
__device__ __forceinline__ unsigned f16_over2(unsigned w) {
  if constexpr (sizeof(T) == 2 && !__is_same(T, _Float16)) return 0u;
  return ((w & 0x7c007c00u) + 0x04000400u) & 0x80008000u;
}
template <typename T>
__device__ __forceinline__ unsigned f16_over4(uint2 q) { return f16_over2<T>(q.x) | f16_over2<T>(q.y); }
template <typename T, typename V4>
__device__ __forceinline__ unsigned f16_over_v4(V4 v) { return f16_over4<T>(__builtin_bit_cast(uint2, v)); }
__device__ __forceinline__ void f16_guard_commit(unsigned* flag, unsigned bad, unsigned site) {
  if (flag != nullptr && bad != 0u) atomicOr(flag, site);
}

__device__ __forceinline__ float silu_f(float v) {
  // x * sigmoid(x); exp/rcp are the hardware transcendental forms (<= 1 ulp-ish), far inside 1e-3.
  return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
}

// exact (erf) GELU of nn.GELU's default: 0.5 v (1 + erf(v / sqrt 2)). erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7): a rational
// argument, a degree-5 polynomial and one hardware exp — a third of libm erff's instructions, which made the GEMM epilogues that apply it
// (fc1 of the ViT / Swin MLPs) compute-bound. The complement y = 1 - erf(|x|) is used directly on the negative side, so the tail keeps
// its relative accuracy.
__device__ __forceinline__ float gelu_erf_f(float v) {
  const float x = fabsf(v) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float y = p * t * __expf(-x * x);          // 1 - erf(|x|)
  return 0.5f * v * (v >= 0.f ? 2.0f - y : y);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum for blockDim.x == 256 (4 waves). `red` is >= 4 floats of LDS. Result valid in all threads.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

}  // namespace stedm
