// Achievable dense MFMA rates of this MI355X under sustained load (no memory traffic): the ceiling the prefill GEMM
// (int8 16x16x64) and the prefill attention (fp16 16x16x32) can be priced against besides the datasheet peaks
// (5 POP/s, 2.5 PFLOP/s).  The MFMAs are inline asm with in-place VGPR accumulators: written with the builtin, hipcc
// moved the accumulators through AGPRs with dozens of v_accvgpr copies per iteration and the loop ran at half rate.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256) void k_i8(int* out, int iters) {
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)blockIdx.x};
  v4i acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (v4i){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 0x7fffffff) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ((long long*)out)[2] = (long long)(clock64() - c0);          // shader cycles
    ((long long*)out)[3] = (long long)(wall_clock64() - w0);     // 100 MHz ticks
  }
}

__global__ __launch_bounds__(256) void k_f16(float* out, int iters) {
  v8h a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * i); }
  v4f acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (v4f){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  void* out; hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu : {1, 2, 4, 8}) {
    const int grid = 256 * wgs_per_cu;
    for (int iters : {100000}) {
      for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0, 0);
          if (which == 0) hipLaunchKernelGGL(k_i8<1>, dim3(grid), dim3(256), 0, 0, (int*)out, iters);
          else hipLaunchKernelGGL(k_f16, dim3(grid), dim3(256), 0, 0, (float*)out, iters);
          hipEventRecord(e1, 0); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          const double mf = (double)grid * 4 * iters * 8;                     // MFMA instructions
          const double ops = mf * (which == 0 ? 2.0 * 16 * 16 * 64 : 2.0 * 16 * 16 * 32);
          long long hc[4] = {0, 0, 0, 0};
          hipMemcpy(hc, out, 32, hipMemcpyDeviceToHost);
          if (rep == 1 && which == 0) printf("    shader clock under load: %.0f MHz\n", hc[3] ? 100.0 * hc[2] / hc[3] : 0.0);
          if (rep == 1)
            printf("%s  %d WG/CU (%d waves/SIMD)  iters %6d : %8.3f ms  %7.1f T%s/s  (%.2f cycles per MFMA per SIMD at 2.4 GHz)\n",
                   which == 0 ? "i8  16x16x64" : "f16 16x16x32", wgs_per_cu, wgs_per_cu, iters, ms, ops / ms * 1e-9,
                   which == 0 ? "OP" : "FLOP", ms * 1e-3 * 2.4e9 / ((double)wgs_per_cu * iters * 8));
        }
      }
    }
  }
  return 0;
}
