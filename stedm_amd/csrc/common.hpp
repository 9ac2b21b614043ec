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

__device__ __forceinline__ float silu_f(float v) {
  // x * sigmoid(x); exp/rcp are the hardware transcendental forms (<= 1 ulp-ish), far inside 1e-3.
  return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
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
