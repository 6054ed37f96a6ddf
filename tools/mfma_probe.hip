// Verifies on real gfx950 hardware the operand/result layout of v_mfma_i32_16x16x64_i8 that
// omniserve_amd/csrc/qgemm_kernel.h relies on:
//   A lane l byte j  = A[row l&15][k = (l>>4)*16 + j]
//   B lane l byte j  = B[k = (l>>4)*16 + j][col l&15]
//   D lane l reg r   = D[row (l>>4)*4 + r][col l&15]
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 tools/mfma_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k(const v4i* a, const v4i* b, v4i* d) {
  v4i acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  d[threadIdx.x] = acc;
}

int main() {
  int8_t A[16][64], B[64][16];
  int32_t D[16][16];
  srand(1);
  for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 64; ++kk) A[i][kk] = (int8_t)(rand() % 255 - 127);
  for (int kk = 0; kk < 64; ++kk) for (int j = 0; j < 16; ++j) B[kk][j] = (int8_t)(rand() % 255 - 127);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    int s = 0;
    for (int kk = 0; kk < 64; ++kk) s += (int)A[i][kk] * (int)B[kk][j];
    D[i][j] = s;
  }
  int8_t ha[64][16], hb[64][16];
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) {
    ha[l][j] = A[l & 15][(l >> 4) * 16 + j];
    hb[l][j] = B[(l >> 4) * 16 + j][l & 15];
  }
  v4i *da, *db, *dd;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, 64 * 16);
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
  hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
  int32_t hd[64][4];
  hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r)
    if (hd[l][r] != D[(l >> 4) * 4 + r][l & 15]) ++bad;
  printf("mfma_i32_16x16x64_i8 layout hypothesis: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  if (bad) {  // dump enough to re-derive the layout
    for (int l = 0; l < 64; l += 5) printf("lane %d: %d %d %d %d\n", l, hd[l][0], hd[l][1], hd[l][2], hd[l][3]);
    // locate each produced value in the reference D
    for (int l = 0; l < 8; ++l) for (int r = 0; r < 4; ++r)
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j)
        if (D[i][j] == hd[l][r]) printf("  lane %d reg %d == D[%d][%d]\n", l, r, i, j);
  }
  return bad != 0;
}
