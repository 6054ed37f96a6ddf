// How fast does a CU pull 16-B-per-lane loads out of L2 / L1 as a function of the lane -> address map?
// Each wave issues global_load_dwordx4 over a [rows][row_bytes] int8 matrix (the W8A8 weight layout, row pitch 4096 B):
//   map 0  contiguous:        lane l -> byte 16 * l of a 1-KiB run                       (8 lines of 128 B per instruction)
//   map 1  MFMA operand map:  lane l -> row l & 15, 16-B piece l >> 4 of a 64-B k-step   (16 half lines; quads span 4 rows)
//   map 2  row-coalesced:     lane l -> row l >> 2, piece l & 3                          (16 half lines; a quad = 64 B of one row)
//   map 3  row-coalesced 128: lane l -> row l >> 3, piece l & 7 (two k-steps)            (8 whole lines)
// The footprint per workgroup (64 rows x 4 KiB = 256 KiB) streams from L2; the loop walks k so that every byte is used once,
// like the GEMM / GEMV weight stream.   hipcc --offload-arch=gfx950 -O3 tools/l1_pattern_probe.hip -o /tmp/l1p && /tmp/l1p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int PITCH = 4096;

template <int MAP>
__global__ __launch_bounds__(256) void probe(const uint8_t* W, int* out, int rows_total, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // a wave owns 64 rows (MAP 0 treats them as one flat 256-KiB run)
  const size_t wave_base = ((size_t)(blockIdx.x * 4 + wave) * 64 % rows_total) * PITCH;
  v4i acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll 4
    for (int k = 0; k < PITCH; k += 64) {          // one 64-B k-step of 64 rows = 4 KiB = 4 instructions
      v4i v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        size_t off;
        if (MAP == 0) off = wave_base + (size_t)(k / 64) * 4096 + j * 1024 + lane * 16;
        else if (MAP == 1) off = wave_base + (size_t)(j * 16 + (lane & 15)) * PITCH + k + (lane >> 4) * 16;
        else if (MAP == 2) off = wave_base + (size_t)(j * 16 + (lane >> 2)) * PITCH + k + (lane & 3) * 16;
        else if (MAP == 3) {   // the packed W4 tile map of the GEMV / GEMM kernels: two 512-B tiles (32 KiB apart) per instruction,
                               // lane l -> tile (l >> 3) & 1, 16-B slot (l & 7) * 4 + (l >> 4) of the tile
          off = wave_base + (size_t)(k / 64) * 4096 + (j >> 1) * 2048 + (j & 1) * 1024 * 0 + ((lane >> 3) & 1) * 131072 +
                (size_t)(j & 1) * 512 + (((lane & 7) * 4 + (lane >> 4)) * 16);
        } else {               // the same two tiles, lane-linear inside each tile (what a transposed packing would allow)
          off = wave_base + (size_t)(k / 64) * 4096 + (j >> 1) * 2048 + ((lane >> 5) & 1) * 131072 + (size_t)(j & 1) * 512 +
                (lane & 31) * 16;
        }
        v[j] = *reinterpret_cast<const v4i*>(W + off);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc ^= v[j];
    }
  }
  if (acc[0] == 0x12345678) out[threadIdx.x] = acc[1] ^ acc[2] ^ acc[3];
}

template <int MAP>
static void run(const uint8_t* W, int* out, int rows, const char* name) {
  const int blocks = 512, iters = 8;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(probe<MAP>, dim3(blocks), dim3(256), 0, 0, W, out, rows, 1);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(probe<MAP>, dim3(blocks), dim3(256), 0, 0, W, out, rows, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)blocks * 4 * 64 * PITCH * iters;
  printf("%-34s %8.3f ms  %7.2f TB/s (%5.1f B/clk/CU at 2.4 GHz)\n", name, ms, bytes / ms * 1e-9, bytes / ms * 1e-6 / 256 / 2.4e3);
}

int main() {
  const int rows = 512 * 4 * 64 / 8;     // 64 MiB: the 8 iterations re-read it (L2 + MALL resident after the first pass)
  uint8_t* W; int* out;
  hipMalloc(&W, (size_t)rows * PITCH + (1 << 20)); hipMalloc(&out, 4096);   // + slack: maps 3 / 4 reach 128 KiB past a wave's rows
  hipMemset(W, 1, (size_t)rows * PITCH);
  run<0>(W, out, rows, "contiguous 1 KiB per instruction");
  run<1>(W, out, rows, "MFMA map (row = lane & 15)");
  run<2>(W, out, rows, "row-coalesced (row = lane >> 2)");
  run<3>(W, out, rows, "packed W4 tile map (2 x 512 B)");
  run<4>(W, out, rows, "2 x 512 B, lane-linear");
  return 0;
}
