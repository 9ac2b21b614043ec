// GroupNorm statistics -> per-(sample, channel) scale/shift, over the virtual concat [x1 | x2].
// One block per (sample, group): two passes over its cpg x HW slice (second pass hits L1/L2),
// so the variance is the centred, biased form nn.GroupNorm computes (util.py:214-216).
#include "common.hpp"
using namespace stedm;

struct GnArgs {
  const float* x1;
  const float* x2;
  int c1, c2, bmod, groups, HW;
  const float* gamma;
  const float* beta;
  float eps;
  float* scale;
  float* shift;
};

template <bool VEC4>
__global__ void __launch_bounds__(256) gn_scale_shift_kernel(GnArgs a) {
  __shared__ float red[4];
  const int b = blockIdx.x / a.groups, g = blockIdx.x % a.groups;
  const int C = a.c1 + a.c2;
  const int cpg = C / a.groups;
  const int cbeg = g * cpg;
  const float* p1 = a.x1 + (long)b * a.HW * a.c1;
  const float* p2 = a.x2 ? a.x2 + (long)(a.bmod > 0 ? b % a.bmod : b) * a.HW * a.c2 : nullptr;
  const int n = cpg * a.HW;

  auto load1 = [&](int e) -> float {
    const int pix = e / cpg, c = cbeg + (e - pix * cpg);
    return c < a.c1 ? p1[(long)pix * a.c1 + c] : p2[(long)pix * a.c2 + (c - a.c1)];
  };
  auto load4 = [&](int e4) -> float4 {  // e4 indexes quads; cpg % 4 == 0 and c1 % 4 == 0
    const int qpg = cpg >> 2;
    const int pix = e4 / qpg, c = cbeg + ((e4 - pix * qpg) << 2);
    return c < a.c1 ? *reinterpret_cast<const float4*>(p1 + (long)pix * a.c1 + c)
                    : *reinterpret_cast<const float4*>(p2 + (long)pix * a.c2 + (c - a.c1));
  };

  float s = 0.f;
  if (VEC4) {
    for (int e = threadIdx.x; e < (n >> 2); e += 256) {
      const float4 v = load4(e);
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (int e = threadIdx.x; e < n; e += 256) s += load1(e);
  }
  const float mean = block_sum_256(s, red) / (float)n;
  float q = 0.f;
  if (VEC4) {
    for (int e = threadIdx.x; e < (n >> 2); e += 256) {
      const float4 v = load4(e);
      const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
      q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  } else {
    for (int e = threadIdx.x; e < n; e += 256) {
      const float d = load1(e) - mean;
      q += d * d;
    }
  }
  const float var = block_sum_256(q, red) / (float)n;
  const float rstd = 1.0f / sqrtf(var + a.eps);
  for (int c = threadIdx.x; c < cpg; c += 256) {
    const float gm = a.gamma[cbeg + c] * rstd;
    a.scale[(long)b * C + cbeg + c] = gm;
    a.shift[(long)b * C + cbeg + c] = a.beta[cbeg + c] - mean * gm;
  }
}

extern "C" int stedm_gn_scale_shift(const float* x1, int c1, const float* x2, int c2, int x2_bmod, const float* gamma,
                                    const float* beta, float eps, int groups, int B, int HW, float* scale, float* shift,
                                    void* stream) {
  STEDM_CHECK_ARG(x1 && gamma && beta && scale && shift, "gn_scale_shift: null pointer");
  STEDM_CHECK_ARG((x2 != nullptr) == (c2 > 0), "gn_scale_shift: x2/c2 mismatch");
  const int C = c1 + c2;
  STEDM_CHECK_ARG(groups > 0 && C % groups == 0, "gn_scale_shift: C=%d not divisible by groups=%d", C, groups);
  STEDM_CHECK_ARG(B > 0 && HW > 0, "gn_scale_shift: bad B/HW");
  GnArgs a{x1, x2, c1, c2, x2_bmod, groups, HW, gamma, beta, eps, scale, shift};
  const int cpg = C / groups;
  const bool vec = (cpg % 4 == 0) && (c1 % 4 == 0) && (c2 % 4 == 0);
  if (vec)
    gn_scale_shift_kernel<true><<<B * groups, 256, 0, as_stream(stream)>>>(a);
  else
    gn_scale_shift_kernel<false><<<B * groups, 256, 0, as_stream(stream)>>>(a);
  STEDM_LAUNCH_CHECK();
  return 0;
}
