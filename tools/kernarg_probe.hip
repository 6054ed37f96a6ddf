// Does kernarg preloading (hipcc -mllvm -amdgpu-kernarg-preload-count=N: the CP delivers the first kernel arguments
// in SGPRs at wave launch instead of the wave fetching them with s_load from the kernarg segment) shorten a chain of
// small dependent kernels on this GPU / firmware?  A hipGraph of CHAIN dependent kernels, each with its own argument
// block, each doing one dependent global load + store (16 workgroups): microseconds per kernel, for argument lists
// passed as plain scalars (preloadable) and as one struct (not preloadable).
//   hipcc --offload-arch=gfx950 -O3 tools/kernarg_probe.hip -o tune_libs/kernarg_probe_plain
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 tools/kernarg_probe.hip -o tune_libs/kernarg_probe_preload
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Args { const float* in; float* out; int n; float s; int pad[8]; };
__global__ __launch_bounds__(256) void k_struct(Args a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < a.n) a.out[i] = a.in[i] * a.s + (float)a.pad[3];
}
__global__ __launch_bounds__(256) void k_plain(const float* in, float* out, int n, float s, int p0, int p1, int p2, int p3) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i] * s + (float)p3;
}
__global__ __launch_bounds__(256) void k_noload(float* out, int n) {   // floor: no dependent load
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = 1.0f;
}

int main() {
  const int CHAIN = 256, N = 16 * 256, REPS = 50;
  std::vector<float*> buf(CHAIN + 1);
  for (auto& b : buf) { CK(hipMalloc(&b, N * sizeof(float))); CK(hipMemset(b, 0, N * sizeof(float))); }
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int mode = 0; mode < 3; ++mode) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < CHAIN; ++i) {
      if (mode == 0) { Args a{buf[i], buf[i + 1], N, 1.0f + i, {0, 0, 0, i, 0, 0, 0, 0}}; hipLaunchKernelGGL(k_struct, dim3(16), dim3(256), 0, st, a); }
      else if (mode == 1) hipLaunchKernelGGL(k_plain, dim3(16), dim3(256), 0, st, (const float*)buf[i], buf[i + 1], N, 1.0f + i, 0, 0, 0, i);
      else hipLaunchKernelGGL(k_noload, dim3(16), dim3(256), 0, st, buf[i + 1], N);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s: %.3f us per kernel\n", mode == 0 ? "struct argument (never preloaded)" : mode == 1 ? "plain arguments" : "no dependent load (floor)",
           ms * 1e3 / (REPS * CHAIN));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
