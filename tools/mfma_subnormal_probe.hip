// Do fp16 SUBNORMAL A-operand elements survive v_mfma_f32_16x16x32_f16?  A = raw 4-bit codes as fp16 bit patterns
// (u -> u * 2^-24, and u << 4 -> u * 2^-20), B = 1.0: D must be the exact sums.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_subnormal_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned short v8u __attribute__((ext_vector_type(8)));

__global__ void probe(float* out) {
  const int lane = threadIdx.x;
  v8u a;
  for (int i = 0; i < 8; ++i) a[i] = (unsigned short)(((lane + i) & 15) << (4 * (i & 1)));   // codes at bit 0 or bit 4
  v8h b;
  for (int i = 0; i < 8; ++i) b[i] = (_Float16)((i & 1) ? 0.0625f : 1.0f);                  // 1/16 where the code sits 4 bits up
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a), b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}

int main() {
  float* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // D[i][j]: A row i = lane & 15 (k block lane >> 4); every B column identical -> D[i][*] = sum over 32 k of code(i, k) * 2^-24
  // A lane l (row l & 15, k block l >> 4) holds codes ((l + i) & 15), i = 0..7
  int bad = 0;
  for (int row = 0; row < 16; ++row) {
    double want = 0;
    for (int kb = 0; kb < 4; ++kb) { const int l = row + 16 * kb; for (int i = 0; i < 8; ++i) want += ((l + i) & 15); }
    want *= std::ldexp(1.0, -24);
    // D layout: lane l holds rows 4 * (l >> 4) + r, column l & 15
    const int l = (row / 4) * 16; const float got = h[l * 4 + (row & 3)];
    if (got != (float)want) { ++bad; printf("row %d: got %g want %g\n", row, got, want); }
  }
  printf(bad ? "fp16 subnormal MFMA inputs are NOT exact (%d rows differ)\n" : "fp16 subnormal MFMA inputs come through exactly (%d bad)\n", bad);
  return 0;
}
