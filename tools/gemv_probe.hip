// Ablation probe for the decode GEMV (M=16, per-channel): starts from the pure weight stream and adds
// the real kernel's components one by one (unpack, LDS B-operand reads, MFMA, activation staging,
// slab store), so the component that costs bandwidth shows up.  N=28672, K=4096 (gate_up).
//   hipcc --offload-arch=gfx950 -O3 tools/gemv_probe.hip -o /tmp/gemv_probe && /tmp/gemv_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int RING = 8;
constexpr int MT = 16;

template <int WAVES, int LEVEL>  // LEVEL 0 stream, 1 +unpack, 2 +lds reads, 3 +mfma, 4 +A staging, 5 +slab store
__global__ __launch_bounds__(64 * WAVES) void probe(const uint8_t* W, const int8_t* A, int* out, int N, int K, int nsteps) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ng = blockIdx.x * WAVES + wave;
  const bool active = ng * 64 < N;
  const int lx = (lane >> 3) & 1, lc = lane & 7, le = lane >> 4;
  const uint8_t* base = W + ((size_t)(2 * ng + lx) * (K / 32)) * 512 + (lc * 4 + le) * 16;
  const int k0 = blockIdx.y * nsteps;
  auto ld = [&](int st, int j) -> v4i {
    return __builtin_nontemporal_load(reinterpret_cast<const v4i*>(base + (size_t)((k0 + st) * 2 + j) * 512));
  };
  v4i q[RING][2];
  if (active) {
#pragma unroll
    for (int s = 0; s < RING; ++s) { q[s][0] = ld(s, 0); q[s][1] = ld(s, 1); }
  }
  if (LEVEL >= 4) {
    const int ppr = nsteps * 4, pieces = MT * ppr;
    for (int id0 = tid; id0 < pieces; id0 += 64 * WAVES * 8) {
      uint4 a[8];
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int id = id0 + b * 64 * WAVES; const int m = id / ppr, kk = id - m * ppr;
        a[b] = make_uint4(0, 0, 0, 0);
        if (id < pieces) a[b] = *reinterpret_cast<const uint4*>(A + (size_t)m * K + k0 * 64 + kk * 16);
      }
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int id = id0 + b * 64 * WAVES; if (id >= pieces) continue;
        const int m = id / ppr, kk = id - m * ppr; const int kp = kk >> 2, tp = (kk >> 1) & 1, d = kk & 1;
        uint8_t* dst = &lds[(kp * MT + m) * 64 + tp * 8 + d * 4];
        *reinterpret_cast<uint32_t*>(dst + 0) = a[b].x; *reinterpret_cast<uint32_t*>(dst + 16) = a[b].y;
        *reinterpret_cast<uint32_t*>(dst + 32) = a[b].z; *reinterpret_cast<uint32_t*>(dst + 48) = a[b].w;
      }
    }
    __syncthreads();
  }
  if (!active) return;
  v4i acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
  auto step = [&](const v4i (&w)[2], int st) {
    if (LEVEL == 0) { acc[0] ^= w[0] ^ w[1]; return; }
    const uint32_t d[2][4] = {{(uint32_t)w[0][0], (uint32_t)w[0][2], (uint32_t)w[1][0], (uint32_t)w[1][2]},
                              {(uint32_t)w[0][1], (uint32_t)w[0][3], (uint32_t)w[1][1], (uint32_t)w[1][3]}};
    v4i wa[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        uint32_t u[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) u[x] = (d[b][x] >> (4 * a)) & 0x0F0F0F0Fu;
        wa[a * 2 + b] = (v4i){(int)u[0], (int)u[1], (int)u[2], (int)u[3]};
      }
    v4i bf = {1, 2, 3, 4};
    if (LEVEL >= 2) bf = *reinterpret_cast<const v4i*>(lds + ((st * MT + (lane & 15)) * 4 + (lane >> 4)) * 16);
    if (LEVEL >= 3) {
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) acc[ab] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[ab], bf, acc[ab], 0, 0, 0);
    } else {
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) acc[ab] ^= wa[ab] ^ bf;
    }
  };
  const int rounds = nsteps / RING;
  for (int r = 0; r + 1 < rounds; ++r) {
#pragma unroll
    for (int s = 0; s < RING; ++s) {
      v4i w[2] = {q[s][0], q[s][1]};
      q[s][0] = ld((r + 1) * RING + s, 0); q[s][1] = ld((r + 1) * RING + s, 1);
      step(w, r * RING + s);
    }
  }
#pragma unroll
  for (int s = 0; s < RING; ++s) { v4i w[2] = {q[s][0], q[s][1]}; step(w, (rounds - 1) * RING + s); }
  if (LEVEL >= 5) {
    const int m = lane & 15, i0 = (lane >> 4) * 4;
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
      const int n = ng * 64 + (i0 >> 3) * 32 + ab * 8 + (i0 & 7);
      *reinterpret_cast<v4i*>(out + ((size_t)blockIdx.y * 16 + m) * N + n) = acc[ab];
    }
  } else {
    v4i t = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if ((t[0] ^ t[1] ^ t[2] ^ t[3]) == 0x12345678) out[ng] = 1;
  }
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


// ---- v3 structure: activations staged one ring round ahead; WAVES waves share the staged tile ------
template <bool PIN, int WAVES, int RG = 8, bool DMA = false>
__global__ __launch_bounds__(64 * WAVES, 1) void probe_round(const uint8_t* W, const int8_t* A, int* out, int N, int K, int nsteps) {
  constexpr int RING = RG; constexpr int RK = RING * 64, NT = 64 * WAVES, APT = (MT * RK / 16) / NT;
  __shared__ __attribute__((aligned(16))) uint8_t lds[2][MT * RK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ng = blockIdx.x * WAVES + wave;
  const int lx = (lane >> 3) & 1, lc = lane & 7, le = lane >> 4;
  const uint8_t* base = W + ((size_t)(2 * ng + lx) * (K / 32)) * 512 + (lc * 4 + le) * 16;
  const int k0 = blockIdx.y * nsteps;
  auto ld = [&](int st, int j) -> v4i {
    return __builtin_nontemporal_load(reinterpret_cast<const v4i*>(base + (size_t)((k0 + st) * 2 + j) * 512));
  };
  uint4 areg[APT];
  auto dma_a = [&](int kr, int buf) {   // LDS-DMA: 256 B (4 rows x 64 B of one k-pair) per instruction
#pragma unroll
    for (int kp = 0; kp < RING; ++kp)
#pragma unroll
      for (int mq = 0; mq < MT / 4; ++mq) {
        const int m = mq * 4 + (lane >> 4);
        const int e = (lane >> 2) & 3, tp = (lane >> 1) & 1, d = lane & 1;
        const int8_t* src = A + (size_t)m * K + kr + kp * 64 + tp * 32 + d * 16 + e * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(&lds[buf][(kp * MT + mq * 4) * 64]), 4, 0, 0);
      }
  };
  auto load_a = [&](int kr) {
#pragma unroll
    for (int j = 0; j < APT; ++j) {
      const int id = tid + j * NT; const int m = id / (RK / 16), kk = id % (RK / 16);
      areg[j] = *reinterpret_cast<const uint4*>(A + (size_t)m * K + kr + kk * 16);
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int j = 0; j < APT; ++j) {
      const int id = tid + j * NT; const int m = id / (RK / 16), kk = id % (RK / 16);
      const int kp = kk >> 2, tp = (kk >> 1) & 1, d = kk & 1;
      uint8_t* dst = &lds[buf][(kp * MT + m) * 64 + tp * 8 + d * 4];
      *reinterpret_cast<uint32_t*>(dst + 0) = areg[j].x; *reinterpret_cast<uint32_t*>(dst + 16) = areg[j].y;
      *reinterpret_cast<uint32_t*>(dst + 32) = areg[j].z; *reinterpret_cast<uint32_t*>(dst + 48) = areg[j].w;
    }
  };
  v4i q[RING][2];
#pragma unroll
  for (int s = 0; s < RING; ++s) { q[s][0] = ld(s, 0); q[s][1] = ld(s, 1); }
  if (DMA) dma_a(k0 * 64, 0); else { load_a(k0 * 64); store_a(0); }
  if (WAVES > 1) __syncthreads();
  v4i acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
  auto step = [&](const v4i (&w)[2], const uint8_t* abuf, int s) {
    const uint32_t d[2][4] = {{(uint32_t)w[0][0], (uint32_t)w[0][2], (uint32_t)w[1][0], (uint32_t)w[1][2]},
                              {(uint32_t)w[0][1], (uint32_t)w[0][3], (uint32_t)w[1][1], (uint32_t)w[1][3]}};
    v4i wa[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        uint32_t u[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) u[x] = (d[b][x] >> (4 * a)) & 0x0F0F0F0Fu;
        wa[a * 2 + b] = (v4i){(int)u[0], (int)u[1], (int)u[2], (int)u[3]};
      }
    const v4i bf = *reinterpret_cast<const v4i*>(abuf + ((s * MT + (lane & 15)) * 4 + (lane >> 4)) * 16);
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) acc[ab] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[ab], bf, acc[ab], 0, 0, 0);
  };
  const int rounds = nsteps / RING;
  for (int r = 0; r + 1 < rounds; ++r) {
    if (DMA) dma_a((k0 + (r + 1) * RING) * 64, (r + 1) & 1); else load_a((k0 + (r + 1) * RING) * 64);
    const uint8_t* abuf = lds[r & 1];
#pragma unroll
    for (int s = 0; s < RING; ++s) {
      v4i w[2] = {q[s][0], q[s][1]};
      q[s][0] = ld((r + 1) * RING + s, 0); q[s][1] = ld((r + 1) * RING + s, 1);
      step(w, abuf, s);
      if (PIN) __builtin_amdgcn_sched_barrier(0);
    }
    if (!DMA) store_a((r + 1) & 1);
    if (WAVES > 1) __syncthreads();
  }
  {
    const uint8_t* abuf = lds[(rounds - 1) & 1];
#pragma unroll
    for (int s = 0; s < RING; ++s) { v4i w[2] = {q[s][0], q[s][1]}; step(w, abuf, s); if (PIN) __builtin_amdgcn_sched_barrier(0); }
  }
  const int m = lane & 15, i0 = (lane >> 4) * 4;
#pragma unroll
  for (int ab = 0; ab < 4; ++ab) {
    const int n = ng * 64 + (i0 >> 3) * 32 + ab * 8 + (i0 & 7);
    if (gridDim.y > 1) {
      *reinterpret_cast<v4i*>(out + ((size_t)blockIdx.y * 16 + m) * N + n) = acc[ab];
    } else {
      int2 o = {acc[ab][0] ^ acc[ab][1], acc[ab][2] ^ acc[ab][3]};
      *reinterpret_cast<int2*>(reinterpret_cast<char*>(out) + ((size_t)m * N + n) * 2) = o;
    }
  }
}

template <bool PIN, int WAVES, int RG = 8, bool DMA = false>
void run_round(const uint8_t* W, const int8_t* A, int* out, int N, int K, size_t bytes, int copies) {
  for (int sk : {1, 2, 4}) {
    const int nsteps = K / 64 / sk;
    dim3 grid(N / 64 / WAVES, sk);
    float us = time_us([&](int i) { hipLaunchKernelGGL((probe_round<PIN, WAVES, RG, DMA>), grid, dim3(64 * WAVES), 0, 0,
                                    W + bytes * (i % copies), A, out, N, K, nsteps); }, 24);
    printf("round-staged pin=%d waves=%d ring=%d dma=%d sk=%d : %7.2f us  %7.1f GB/s\n", (int)PIN, WAVES, RG, (int)DMA, sk, us, bytes / us / 1e3);
  }
}

template <int WAVES, int LEVEL>
void run(const uint8_t* W, const int8_t* A, int* out, int N, int K, size_t bytes, int copies) {
  for (int sk : {1, 2, 4, 8}) {
    const int nsteps = K / 64 / sk;
    dim3 grid((N / 64 + WAVES - 1) / WAVES, sk);
    const size_t lds = (LEVEL >= 2) ? (size_t)MT * nsteps * 64 : 0;
    hipFuncSetAttribute((const void*)probe<WAVES, LEVEL>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    float us = time_us([&](int i) { hipLaunchKernelGGL((probe<WAVES, LEVEL>), grid, dim3(64 * WAVES), lds, 0,
                                    W + bytes * (i % copies), A, out, N, K, nsteps); }, 24);
    printf("waves=%d level=%d sk=%d : %7.2f us  %7.1f GB/s\n", WAVES, LEVEL, sk, us, bytes / us / 1e3);
  }
}

int main() {
  const int N = 28672, K = 4096;
  const size_t bytes = (size_t)N * K / 2;
  const int copies = 12;
  uint8_t* W; int8_t* A; int* out;
  hipMalloc(&W, bytes * copies); hipMalloc(&A, 16 * K); hipMalloc(&out, (size_t)8 * 16 * N * 4);
  hipMemset(W, 0x5a, bytes * copies); hipMemset(A, 1, 16 * K);
  printf("levels: 0 stream, 1 +unpack, 2 +lds B reads, 3 +mfma, 4 +A staging, 5 +slab store\n");
  run_round<false, 1>(W, A, out, N, K, bytes, copies);
  run_round<false, 1, 16>(W, A, out, N, K, bytes, copies);
  run_round<false, 1, 4>(W, A, out, N, K, bytes, copies);
  run_round<false, 1, 8, true>(W, A, out, N, K, bytes, copies);
  run_round<false, 1, 16, true>(W, A, out, N, K, bytes, copies);
  run<1, 0>(W, A, out, N, K, bytes, copies); run<1, 1>(W, A, out, N, K, bytes, copies);
  run<1, 3>(W, A, out, N, K, bytes, copies);
  return 0;
}
