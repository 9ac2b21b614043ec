// probe 1 of 2: operand pairing and C/D layout of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands and uniform scales (exact integer data).
// It shows that byte t of lane l of A meets byte t of lane l' of B when l >> 5 == l' >> 5 (same lane half, same byte), that C/D is the 32x32 bf16
// layout and that the scale bytes are E8M0 (127 = 1). It can NOT tell which k a (lane half, byte) is, nor which bytes a lane's scale covers:
// probe_mx_fp8_scales.hip and the structured-input run in stedm_amd/csrc/attn_fp8.hip's history did (k = 32 (t >> 4) + 16 h + (t & 15);
// the scale of lane (r, h) covers bytes 16 h .. 16 h + 15 of both lanes of row r).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// e4m3 encode of small integers 0..15 exactly: value v -> byte
__host__ __device__ static uint8_t e4m3(float f) {
  if (f == 0.f) return 0;
  int s = f < 0; f = fabsf(f);
  int e; float m = frexpf(f, &e);   // f = m * 2^e, m in [0.5,1)
  // e4m3: value = 2^(E-7) * (1 + M/8), E in 1..15
  int E = e - 1 + 7; int M = (int)roundf((m * 2 - 1) * 8);
  if (M == 8) { M = 0; E++; }
  return (uint8_t)((s << 7) | (E << 3) | M);
}
__global__ void probe(const uint8_t* A, const uint8_t* B, float* C, int sa, int sb, int osa, int osb) {
  // A: [64 lanes][32 bytes], B: same; raw per-lane operands
  const int l = threadIdx.x;
  v8i a, b;
  for (int j = 0; j < 8; ++j) { a[j] = ((const int*)A)[l * 8 + j]; b[j] = ((const int*)B)[l * 8 + j]; }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  for (int e = 0; e < 16; ++e) C[l * 16 + e] = c[e];
}
int main() {
  uint8_t hA[64 * 32], hB[64 * 32]; float hC[64 * 16];
  uint8_t *dA, *dB; float* dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
  // hypothesis: lane l holds row (l & 31), k = 32 * (l >> 5) + byte index
  // Test 1: A[i][k] = (i % 7) + 1 if k == k0 else 0 ; B[k][j] = (j % 5) + 1 if k == k0 else 0 -> C[i][j] = A[i][k0] * B[k0][j]
  int bad_total = 0;
  for (int k0 = 0; k0 < 64; k0 += 7) {
    for (int l = 0; l < 64; ++l) for (int t = 0; t < 32; ++t) {
      const int k = 32 * (l >> 5) + t, r = l & 31;
      hA[l * 32 + t] = k == k0 ? e4m3((float)((r % 7) + 1)) : 0;
      hB[l * 32 + t] = k == k0 ? e4m3((float)((r % 5) + 1)) : 0;
    }
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dC, 127, 127, 0, 0);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 16; ++e) {
      const int col = l & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
      const float want = (float)(((row % 7) + 1) * ((col % 5) + 1));
      if (hC[l * 16 + e] != want) { if (bad < 3) printf("k0=%d lane %d e %d got %g want %g\n", k0, l, e, hC[l * 16 + e], want); ++bad; }
    }
    printf("k0=%d mismatches %d\n", k0, bad); bad_total += bad;
  }
  // Test 2: scales: all A = 1 at k in lane-half h's block only -> use scale_a = 128 (2.0) for lanes... scale is a per-lane VGPR value: pass via kernel args uniform first
  for (int l = 0; l < 64; ++l) for (int t = 0; t < 32; ++t) { hA[l * 32 + t] = e4m3(1.f); hB[l * 32 + t] = e4m3(1.f); }
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  int tests[4][2] = {{127, 127}, {128, 127}, {127, 129}, {126, 126}};
  for (auto& t : tests) {
    probe<<<1, 64>>>(dA, dB, dC, t[0], t[1], 0, 0);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    printf("scale_a=%d scale_b=%d -> C[0][0]=%g (64 ones; x 2^(sa-127) x 2^(sb-127) expected)\n", t[0], t[1], hC[0]);
  }
  printf("TOTAL mismatches %d\n", bad_total);
  return 0;
}
