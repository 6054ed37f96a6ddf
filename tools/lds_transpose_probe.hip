// Cost of the ds_write_b128 / ds_read_b128 lane maps of the W8A8 weight-fragment transpose (qgemm_kernel.h), one wave
// per SIMD hammering a private KiB: cycles per instruction for candidate (row r = lane >> 2, piece p = lane & 3) -> slot maps.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_transpose_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int wslot(int map, int lane) {
  const int r = lane >> 2, p = lane & 3;
  switch (map) {
    case 0: return lane;                                   // lane-linear (no transpose): reference cost
    case 1: return p * 16 + (r ^ (4 * p));                 // current
    case 2: return p * 16 + ((r + 4 * p) & 15);            // rotation by 4p
    case 3: return p * 16 + (r ^ p);                       // xor p
    case 4: return p * 16 + (r ^ (5 * p));                 // xor 5p
    case 5: return p * 16 + ((r & 7) * 2 + (r >> 3)) ;     // interleave row halves
    case 6: return p * 16 + ((((r & 7) * 2 + (r >> 3))) ^ (4 * p));
    default: return p * 16 + r;                            // plain transpose (expected: conflicts)
  }
}
__device__ __forceinline__ int rslot(int map, int lane) {   // reader lane l' = 16 p + r wants what (r, p) wrote
  const int p = lane >> 4, r = lane & 15;
  return wslot(map, 4 * r + p);
}

template <int MODE>   // 0 write only, 1 read only, 2 write + read
__global__ void probe(int map, int iters, long long* out, int* sink) {
  __shared__ __attribute__((aligned(16))) uint8_t buf[4][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t* w = buf[wave] + wslot(map, lane) * 16;
  const uint8_t* r = buf[wave] + rslot(map, lane) * 16;
  v4i v = {lane, lane + 1, lane + 2, lane + 3}, acc = {0, 0, 0, 0};
  *reinterpret_cast<v4i*>(buf[wave] + lane * 16) = v;
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE != 1) { *reinterpret_cast<v4i*>(w) = v; }
    if (MODE != 0) { v4i t = *reinterpret_cast<const v4i*>(r); acc += t; }
    if (MODE == 0) v[0] += 1;
    asm volatile("" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (acc[0] == 0x7fffffff) sink[0] = acc[1] + v[0];
}

int main() {
  long long* d; int* sink;
  hipMalloc(&d, 8); hipMalloc(&sink, 4);
  const int iters = 20000;
  const char* names[8] = {"lane-linear", "p*16 + (r ^ 4p)  [current]", "p*16 + (r + 4p) % 16", "p*16 + (r ^ p)", "p*16 + (r ^ 5p)",
                          "p*16 + interleave(r)", "p*16 + (interleave(r) ^ 4p)", "p*16 + r  [plain]"};
  for (int map = 0; map < 8; ++map) {
    double c[3];
    for (int mode = 0; mode < 3; ++mode) {
      long long h = 0;
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, map, iters, d, sink);
        if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, map, iters, d, sink);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, map, iters, d, sink);
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
      }
      c[mode] = (double)h / iters;
    }
    printf("%-32s write %6.1f  read %6.1f  write+read %6.1f  clock64 ticks per iteration (4 waves per CU share the LDS)\n",
           names[map], c[0], c[1], c[2]);
  }
  return 0;
}
