// Where does workgroup i of a 1-D launch run, and when?  3 workgroups of 256 threads fit a CU (50 KiB of LDS each), 2048 workgroups:
// prints (XCC, SE, CU) of the first workgroups in dispatch order and how the three slots of one CU are filled.
//   hipcc --offload-arch=gfx950 -O2 tools/placement_probe.hip -o /tmp/placement_probe && /tmp/placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
  __shared__ unsigned char pad[50 * 1024];
  pad[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(16);
  if (threadIdx.x == 0) {
    out[4 * blockIdx.x + 0] = hw;
    out[4 * blockIdx.x + 1] = xcc;
    out[4 * blockIdx.x + 2] = (unsigned)t0;
    out[4 * blockIdx.x + 3] = pad[7];
  }
}
int main() {
  const int n = 2048;
  unsigned* d;
  hipMalloc(&d, n * 16);
  hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, d, 20);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, d, 20);
  hipDeviceSynchronize();
  std::vector<unsigned> h(4 * n);
  hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> by_cu;
  for (int i = 0; i < n; ++i) {
    const unsigned hw = h[4 * i], xcc = h[4 * i + 1] & 0xF;
    const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    const unsigned key = (xcc << 16) | (se << 8) | (sh << 4) | cu;
    by_cu[key].push_back(i);
    if (i < 24) printf("wg %4d: xcc %u se %u sh %u cu %2u  t0 %u\n", i, xcc, se, sh, cu, h[4 * i + 2]);
  }
  printf("%zu distinct (xcc, se, sh, cu)\n", by_cu.size());
  int shown = 0;
  for (auto& kv : by_cu) {
    if (shown++ >= 6) break;
    printf("cu key %06x:", kv.first);
    for (int i : kv.second) printf(" %d", i);
    printf("\n");
  }
  return 0;
}
