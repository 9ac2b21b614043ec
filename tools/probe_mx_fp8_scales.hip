// probe 2 of 2: which row / column does a lane's E8M0 scale byte apply to in v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 operands)? (row l & 31, one of
// the two 32-wide k blocks: l >> 5.) Part 2 below: which BYTES of the lanes that block is — bytes 16 b .. 16 b + 15 of both lane halves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const int* SA, const int* SB, float* C) {
  const int l = threadIdx.x;
  v8i a, b;
  for (int j = 0; j < 8; ++j) { a[j] = 0x38383838; b[j] = 0x38383838; }   // e4m3 0x38 = 1.0
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, SA[l], 0, SB[l]);
  for (int e = 0; e < 16; ++e) C[l * 16 + e] = c[e];
}
int main() {
  int hSA[64], hSB[64]; float hC[64 * 16];
  int *dSA, *dSB; float* dC;
  (void)hipMalloc(&dSA, sizeof hSA); (void)hipMalloc(&dSB, sizeof hSB); (void)hipMalloc(&dC, sizeof hC);
  // hypothesis: lane l's scale_a scales A[row l & 31][k block l >> 5]; scale_b scales B[k block l >> 5][col l & 31]
  // test A: scale_a of lane l = 127 + (l == L0 ? 3 : 0): expected C[row][col] = 32 * (1 + ... ) -> row (L0 & 31) gets 32 * 8 + 32 = 288, others 64
  for (int L0 : {0, 5, 37, 63}) {
    for (int l = 0; l < 64; ++l) { hSA[l] = 127 + (l == L0 ? 3 : 0); hSB[l] = 127; }
    (void)hipMemcpy(dSA, hSA, sizeof hSA, hipMemcpyHostToDevice); (void)hipMemcpy(dSB, hSB, sizeof hSB, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dSA, dSB, dC);
    (void)hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    printf("scale_a bump at lane %d: rows != 64:", L0);
    for (int row = 0; row < 32; ++row) {
      // C[row][col 0]: lane 0 or 32 holds col 0; row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
      const int hh = (row >> 2) & 1, e = (row & 3) + 4 * (row >> 3);
      const float v = hC[(32 * hh) * 16 + e];
      if (v != 64.f) printf(" row %d = %g", row, v);
    }
    printf("\n");
    for (int l = 0; l < 64; ++l) { hSB[l] = 127 + (l == L0 ? 3 : 0); hSA[l] = 127; }
    (void)hipMemcpy(dSA, hSA, sizeof hSA, hipMemcpyHostToDevice); (void)hipMemcpy(dSB, hSB, sizeof hSB, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dSA, dSB, dC);
    (void)hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    printf("scale_b bump at lane %d: cols != 64 (row 0):", L0);
    for (int col = 0; col < 32; ++col) { const float v = hC[col * 16 + 0]; if (v != 64.f) printf(" col %d = %g", col, v); }
    printf("\n");
  }
  // part 2: A = one-hot byte (lane LA, byte TA), B = all ones, scale_a bumped (x 8) on the lanes of half HB only: the result row is 8 iff
  // byte TA of lane half (LA >> 5) lies in scale block HB
  return 0;
}
