// Empirically determines the data movement of ds_read_b64_tr_b16 on gfx950: every LDS half holds its
// own index, lane l passes byte address addr(l); prints what each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __fp16 v4h __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(int mode, short* out) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int off;  // in halves
  if (mode == 0) off = l * 4;                                   // contiguous 8 B per lane
  else if (mode == 1) off = (l & 15) * 4 + (l >> 4) * 256;      // 16-lane groups 512 B apart
  else off = ((l & 15) >> 2) * 64 + (l & 3) * 4 + (l >> 4) * 16; // [4 rows][stride 64 halves], 16-col blocks
  v4h r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) v4h*)(__attribute__((address_space(3))) void*)(lds + off));
  v4s s = __builtin_bit_cast(v4s, r);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = s[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
