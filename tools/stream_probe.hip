// Bandwidth probe for the decode-GEMV access pattern on MI355X: how fast can the weight stream alone
// go (no unpack / MFMA), as a function of cache policy (nt), start rotation, waves per workgroup and
// K split?  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int RING = 8;

// N x K packed int4 matrix = (N/32) tile rows of K*16 bytes.  A wave owns two tile rows and the k-steps
// [ks*nsteps, (ks+1)*nsteps) of 64 k (= 2 x 512 B per tile row per step).
template <bool NT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void stream_kernel(const uint8_t* W, int* out, int N, int K, int nsteps, int rot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ng = blockIdx.x * WAVES + wave;
  if (ng * 64 >= N) return;
  const int lx = (lane >> 3) & 1, lc = lane & 7, le = lane >> 4;
  const uint8_t* base = W + ((size_t)(2 * ng + lx) * (K / 32)) * 512 + (lc * 4 + le) * 16;
  const int k0 = blockIdx.y * nsteps;
  const int r0 = rot ? (ng * 5) % nsteps : 0;
  auto ld = [&](int st, int j) -> v4i {
    int s = st + r0; if (s >= nsteps) s -= nsteps;
    const v4i* p = reinterpret_cast<const v4i*>(base + (size_t)((k0 + s) * 2 + j) * 512);
    if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
  };
  v4i q[RING][2];
#pragma unroll
  for (int s = 0; s < RING; ++s) { q[s][0] = ld(s, 0); q[s][1] = ld(s, 1); }
  v4i acc = {0, 0, 0, 0};
  const int rounds = nsteps / RING;
  for (int r = 0; r + 1 < rounds; ++r) {
#pragma unroll
    for (int s = 0; s < RING; ++s) {
      acc ^= q[s][0] ^ q[s][1];
      q[s][0] = ld((r + 1) * RING + s, 0);
      q[s][1] = ld((r + 1) * RING + s, 1);
    }
  }
#pragma unroll
  for (int s = 0; s < RING; ++s) acc ^= q[s][0] ^ q[s][1];
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) out[ng] = 1;
}

// reference: plain contiguous grid-stride 16 B/lane read of the same bytes
__global__ __launch_bounds__(256) void linear_kernel(const v4i* W, int* out, size_t n16) {
  v4i acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    acc ^= __builtin_nontemporal_load(W + i);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) out[0] = 1;
}

template <typename F>
float time_us(F launch, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 4; ++i) launch(i);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) launch(i);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / iters;
}

int main() {
  const int N = 28672, K = 4096;
  const size_t bytes = (size_t)N * K / 2;
  const int copies = 12;
  uint8_t* W; int* out;
  hipMalloc(&W, bytes * copies); hipMalloc(&out, 1 << 20);
  hipMemset(W, 0x5a, bytes * copies);
  printf("matrix %d x %d int4 = %.1f MB, %d rotating copies\n", N, K, bytes / 1e6, copies);
  {
    float us = time_us([&](int i) { hipLaunchKernelGGL(linear_kernel, dim3(2048), dim3(256), 0, 0,
                                    (const v4i*)(W + bytes * (i % copies)), out, bytes / 16); }, 24);
    printf("linear nt grid-stride            : %7.2f us  %7.1f GB/s\n", us, bytes / us / 1e3);
  }
  for (int nt = 0; nt < 2; ++nt)
    for (int rot = 0; rot < 2; ++rot)
      for (int waves : {1, 4})
        for (int sk : {1, 2, 4, 8}) {
          const int nsteps = K / 64 / sk;
          dim3 grid((N / 64 + waves - 1) / waves, sk);
          auto launch = [&](int i) {
            const uint8_t* w = W + bytes * (i % copies);
            if (nt) { if (waves == 1) hipLaunchKernelGGL((stream_kernel<true, 1>), grid, dim3(64), 0, 0, w, out, N, K, nsteps, rot);
                      else hipLaunchKernelGGL((stream_kernel<true, 4>), grid, dim3(256), 0, 0, w, out, N, K, nsteps, rot); }
            else    { if (waves == 1) hipLaunchKernelGGL((stream_kernel<false, 1>), grid, dim3(64), 0, 0, w, out, N, K, nsteps, rot);
                      else hipLaunchKernelGGL((stream_kernel<false, 4>), grid, dim3(256), 0, 0, w, out, N, K, nsteps, rot); }
          };
          float us = time_us(launch, 24);
          printf("ring nt=%d rot=%d waves=%d sk=%d     : %7.2f us  %7.1f GB/s\n", nt, rot, waves, sk, us, bytes / us / 1e3);
        }
  return 0;
}
