// Weight stream of the M = 16 gate_up GEMV (28672 x 4096 int4 = 58.7 MB, cold, rotating copies): VGPR ring (what the GEMV
// does) against an LDS-DMA ring (global_load_lds_dwordx4 into a wave-private LDS ring, read back with ds_read_b128).
// Same bytes per wave and the same (channel group, K part) ownership.  Question: does the DMA path lift the ~10 B/clk/CU
// that vector loads missing to HBM get on this chip (MI355X_MICROARCH.md; tools/gemv_balance.py: 5.8 us per 131-KiB workgroup
// and CU)?     hipcc --offload-arch=gfx950 -O3 tools/stream_dma_probe.hip -o /tmp/sdp && /tmp/sdp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));

// COAL: lane l asks for bytes 16 l .. of a 1-KiB piece (piece j of a k-step = tiles 2 s, 2 s + 1 of tile row 2 ng + j): every lane quad
// reads 64 ascending bytes.  !COAL: the MFMA operand's own lane map over the packed tile (what the GEMV kernels use): the lanes
// of a quad are 64 B apart (round 5: the address unit takes one COALESCED quad per clock, a scattered one in four).
template <bool NT, int RING, bool COAL = false>
__global__ __launch_bounds__(256) void vgpr_kernel(const uint8_t* W, int* out, int K, int nsteps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ng = blockIdx.x;
  const int lx = (lane >> 3) & 1, lc = lane & 7, le = lane >> 4;
  const uint8_t* base = W + ((size_t)(2 * ng + lx) * (K / 32)) * 512 + (lc * 4 + le) * 16;
  const int k0 = (blockIdx.y * 4 + wave) * nsteps;
  const uint8_t* cbase = W + (size_t)(2 * ng) * (K / 32) * 512 + lane * 16;
  auto ld = [&](int s, int j) -> v4i {
    const v4i* p = COAL ? reinterpret_cast<const v4i*>(cbase + ((size_t)j * (K / 32) + (size_t)(k0 + s) * 2) * 512)
                        : reinterpret_cast<const v4i*>(base + (size_t)((k0 + s) * 2 + j) * 512);
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

// DMA ring: PIECES 1-KiB pieces in flight per wave (two pieces = one 64-k step of the wave's two tile rows)
template <bool NT, int PIECES, bool COAL = false>
__global__ __launch_bounds__(256) void dma_kernel(const uint8_t* W, int* out, int K, int nsteps) {
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ng = blockIdx.x;
  const int lx = (lane >> 3) & 1, lc = lane & 7, le = lane >> 4;
  const uint8_t* base = W + ((size_t)(2 * ng + lx) * (K / 32)) * 512 + (lc * 4 + le) * 16;
  const int k0 = (blockIdx.y * 4 + wave) * nsteps;
  const uint32_t ring = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)smem + wave * PIECES * 1024;
  const int npieces = nsteps * 2;
  auto issue = [&](int pc) {      // piece pc of this wave -> ring slot pc % PIECES
    const uint8_t* cb = W + (size_t)(2 * ng) * (K / 32) * 512 + lane * 16;
    const uint8_t* src = COAL ? cb + ((size_t)(pc & 1) * (K / 32) + (size_t)(k0 + (pc >> 1)) * 2) * 512 : base + (size_t)(k0 * 2 + pc) * 512;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(ring + (uint32_t)(pc % PIECES) * 1024);
    uint32_t keep;
    if constexpr (NT)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    else
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  };
#pragma unroll
  for (int pc = 0; pc < PIECES; ++pc) issue(pc);
  v4i acc = {0, 0, 0, 0};
  const uint8_t* rd = smem + wave * PIECES * 1024 + lane * 16;
  for (int pc = 0; pc < npieces; pc += 2) {       // consume two pieces (one k-step), refill their slots
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES - 2) : "memory");
    const int slot = pc % PIECES;
    const v4i a = *reinterpret_cast<const v4i*>(rd + slot * 1024);
    const v4i b = *reinterpret_cast<const v4i*>(rd + (slot + 1) * 1024);
    acc ^= a ^ b;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the slot is read before it is refilled
    if (pc + PIECES < npieces) { issue(pc + PIECES); issue(pc + PIECES + 1); }
    else { asm volatile("s_nop 0" ::: "memory"); }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) out[ng] = 1;
}
// (the tail leaves fewer than PIECES - 2 pieces outstanding: the counted wait then over-waits, which is what a tail does)

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
  printf("matrix %d x %d int4 = %.1f MB, %d rotating copies; 448 workgroups x 4 waves x sk\n", N, K, bytes / 1e6, copies);
  for (int sk : {1, 2}) {
    const int nsteps = K / 64 / 4 / sk;
    dim3 grid(N / 64, sk);
#define RUN(name, kern, lds)                                                                                         \
    { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                        \
      float us = time_us([&](int i) { hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, W + bytes * (i % copies), out, K, nsteps); }, 36); \
      printf("sk=%d %-34s: %7.2f us  %7.1f GB/s\n", sk, name, us, bytes / us / 1e3); }
    RUN("vgpr ring 8 steps nt", (vgpr_kernel<true, 8>), 0)
    RUN("vgpr ring 8 steps nt COALESCED", (vgpr_kernel<true, 8, true>), 0)
    RUN("vgpr ring 8 steps COALESCED", (vgpr_kernel<false, 8, true>), 0)
    RUN("vgpr ring 8 steps", (vgpr_kernel<false, 8>), 0)
    if (sk == 1) RUN("vgpr ring 16 steps nt", (vgpr_kernel<true, 16>), 0)
    RUN("dma ring 8 pieces", (dma_kernel<false, 8>), 4 * 8 * 1024)
    RUN("dma ring 8 pieces nt", (dma_kernel<true, 8>), 4 * 8 * 1024)
    RUN("dma ring 16 pieces", (dma_kernel<false, 16>), 4 * 16 * 1024)
    RUN("dma ring 16 pieces nt", (dma_kernel<true, 16>), 4 * 16 * 1024)
    RUN("dma ring 16 pieces nt COALESCED", (dma_kernel<true, 16, true>), 4 * 16 * 1024)
    RUN("dma ring 8 pieces nt COALESCED", (dma_kernel<true, 8, true>), 4 * 8 * 1024)
    if (sk == 1) RUN("dma ring 32 pieces nt", (dma_kernel<true, 32>), 4 * 32 * 1024)
  }
  return 0;
}
