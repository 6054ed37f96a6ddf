// Does global_load_lds_dwordx4 reach LDS addresses beyond 64 KiB on gfx950 (M0 carries the wave's LDS base)?
// hipcc --offload-arch=gfx950 -O2 tools/lds_dma_hi_probe.hip -o /tmp/lds_dma_hi_probe && /tmp/lds_dma_hi_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(64) void probe(const uint32_t* src, uint32_t* out, int noffs, const uint32_t* offs) {
  __shared__ __attribute__((aligned(1024))) uint8_t smem[144 * 1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 144 * 256; i += 64) reinterpret_cast<uint32_t*>(smem)[i] = 0xDEADBEEFu;
  __syncthreads();
  const uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)smem;
  for (int t = 0; t < noffs; ++t) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane(base + offs[t]);
    const uint32_t voff = (uint32_t)(t * 1024 + lane * 16);
    uint32_t keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "s"(src), "v"(voff), "s"(dst) : "memory");
  }
  __syncthreads();
  for (int t = 0; t < noffs; ++t)
    for (int j = 0; j < 4; ++j) out[(t * 64 + lane) * 4 + j] = reinterpret_cast<uint32_t*>(smem + offs[t])[lane * 4 + j];
}

int main() {
  const std::vector<uint32_t> offs = {0, 32768, 64512, 65536, 66560, 98304, 130048, 146432};
  const int n = (int)offs.size();
  std::vector<uint32_t> h(n * 256);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x1000000u + (uint32_t)i;
  uint32_t *d_src, *d_out, *d_offs;
  hipMalloc(&d_src, h.size() * 4); hipMalloc(&d_out, h.size() * 4); hipMalloc(&d_offs, n * 4);
  hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_offs, offs.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_src, d_out, n, d_offs);
  std::vector<uint32_t> o(h.size());
  hipError_t e = hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost);
  printf("status %s\n", hipGetErrorString(e));
  for (int t = 0; t < n; ++t) {
    int ok = 0;
    for (int i = 0; i < 256; ++i) ok += o[t * 256 + i] == h[t * 256 + i];
    printf("LDS offset %6u: %3d / 256 dwords landed (first %08x)\n", offs[t], ok, o[t * 256]);
  }
  return 0;
}
