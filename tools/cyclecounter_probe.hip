// What does __builtin_readcyclecounter() count on gfx950?  One wave runs N dependent v_add_f32 bracketed by the cycle counter and by the
// 100 MHz wall clock (s_memrealtime).  Measured (idle chip): 2396.8 counter ticks per microsecond -- the shader clock at its 2.4 GHz
// maximum or a constant 2.4 GHz reference (under the power-capped MFMA kernels sclk reads 2.03-2.06 GHz, profiles/r02_f: the stamps
// of tools/*_timeline.py may then overcount shader cycles by up to 1.17x); 36 ticks per loop iteration of one dependent v_add.
//   hipcc --offload-arch=gfx950 -O3 tools/cyclecounter_probe.hip -o /tmp/ccp && /tmp/ccp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned long long* out, float* sink, int n) {
  float x = threadIdx.x;
  unsigned long long c0 = __builtin_readcyclecounter(), w0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x));
  unsigned long long c1 = __builtin_readcyclecounter(), w1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
  sink[threadIdx.x] = x;
}
int main() {
  unsigned long long* d; float* sk; unsigned long long h[2];
  hipMalloc(&d, 64); hipMalloc(&sk, 256);
  for (int rep = 0; rep < 3; ++rep) {
    const int n = 1 << 20;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, sk, n);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("n = %d dependent v_add_f32: %llu counter ticks (%.3f per op), %llu wall ticks of 10 ns -> %.3f ns per op, counter = %.1f MHz\n", n, h[0],
           (double)h[0] / n, h[1], h[1] * 10.0 / n, h[0] / (h[1] * 10.0) * 1e3);
  }
  return 0;
}
