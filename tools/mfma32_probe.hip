// Checks the operand / result layouts of v_mfma_f32_32x32x16_f16 on gfx950 that the 32-row prefill attention assumes:
//   A lane l: row l%32, k = 8*(l/32) + i;  B lane l: column l%32, k = 8*(l/32) + i;
//   D lane l: column l%32, register r: row (r&3) + 8*(r>>2) + 4*(l/32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(const _Float16* A, const _Float16* B, float* D) {   // A [32][16], B [16][32] row-major, D [32][32]
  const int l = threadIdx.x, lo = l & 31, hi = l >> 5;
  v8h a, b;
  for (int i = 0; i < 8; ++i) { a[i] = A[lo * 16 + 8 * hi + i]; b[i] = B[(8 * hi + i) * 32 + lo]; }
  v16f c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + lo] = c[r];
}
int main() {
  _Float16 hA[512], hB[512]; float hD[1024], ref[1024];
  srand(1);
  for (int i = 0; i < 512; ++i) { hA[i] = (_Float16)((rand() % 17) - 8); hB[i] = (_Float16)((rand() % 13) - 6); }
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += (float)hA[m * 16 + kk] * (float)hB[kk * 32 + n]; ref[m * 32 + n] = s; }
  _Float16 *dA, *dB; float* dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 1024; ++i) if (hD[i] != ref[i]) ++bad;
  printf("mfma_f32_32x32x16_f16 layout check: %d mismatches of 1024\n", bad);
  return bad != 0;
}
