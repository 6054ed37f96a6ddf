// Shared device helpers for the OmniServe MI355X (gfx950) hot-path kernels.
// wave = 64 lanes everywhere; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace omni {

typedef _Float16 half_t;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

#define OMNI_OK 0
#define OMNI_EINVAL (-22)
#define OMNI_ENOMEM (-12)
#define OMNI_ELAUNCH (-5)

static inline int omni_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? OMNI_OK : OMNI_ELAUNCH;
}

// Keeps an f32 intermediate as a materialised, once-rounded f32 value.  Without it the backend
// folds `(half)(a * (float)b_half)` into v_fma_mixlo_f16 (one rounding straight to fp16), whereas
// the reference (and oracle/) round to f32 first and to fp16 second; the two differ in ~2^-13 of
// the elements.
__device__ __forceinline__ float rounded_f32(float x) {
  asm volatile("" : "+v"(x));
  return x;
}

// cvt.rni.sat.s8.f32 (reference kernels/csrc/utils.cuh:79-84): round half to
// even, saturate to [-128,127], NaN -> 0.
__device__ __forceinline__ int8_t rni_sat_s8(float x) {
  float r = __builtin_rintf(x);
  r = (r != r) ? 0.0f : r;
  r = __builtin_fminf(__builtin_fmaxf(r, -128.0f), 127.0f);
  return (int8_t)(int)r;
}

// cvt.rni.sat.u8.f32
__device__ __forceinline__ uint32_t rni_sat_u8(float x) {
  float r = __builtin_rintf(x);
  r = (r != r) ? 0.0f : r;
  r = __builtin_fminf(__builtin_fmaxf(r, 0.0f), 255.0f);
  return (uint32_t)(int)r;
}

// ---- reference block-reduction trees (kernels/csrc/reduction_utils.cuh:25-164)
// The reference reduces with a 32-lane xor butterfly (masks 16,8,4,2,1) and then
// a butterfly over the zero-padded 32 warp partials.  On wave64 the same masks
// stay inside each 32-lane half, so the float summation tree is reproduced
// exactly: `virtual warp` = 32-lane half of a wave.
__device__ __forceinline__ float half_wave_butterfly_sum(float v) {
#pragma unroll
  for (int mask = 16; mask > 0; mask >>= 1) v = v + __shfl_xor(v, mask, 64);
  return v;
}
__device__ __forceinline__ float half_wave_butterfly_max(float v) {
#pragma unroll
  for (int mask = 16; mask > 0; mask >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, mask, 64));
  return v;
}

// Block-wide sum with the reference tree.  `red` is LDS scratch of >= 32 floats.
// blockDim.x must be a multiple of 32 and <= 1024.  All threads get the result.
__device__ __forceinline__ float ref_block_sum(float v, float* red) {
  const int lane32 = threadIdx.x & 31;
  const int vwarp = threadIdx.x >> 5;
  const int nwarps = blockDim.x >> 5;
  v = half_wave_butterfly_sum(v);
  __syncthreads();  // protect `red` against a previous use
  if (lane32 == 0) red[vwarp] = v;
  __syncthreads();
  float w = (lane32 < nwarps) ? red[lane32] : 0.0f;
  w = half_wave_butterfly_sum(w);
  return w;
}

__device__ __forceinline__ float ref_block_max(float v, float* red, float pad) {
  const int lane32 = threadIdx.x & 31;
  const int vwarp = threadIdx.x >> 5;
  const int nwarps = blockDim.x >> 5;
  v = half_wave_butterfly_max(v);
  __syncthreads();
  if (lane32 == 0) red[vwarp] = v;
  __syncthreads();
  float w = (lane32 < nwarps) ? red[lane32] : pad;
  w = half_wave_butterfly_max(w);
  return w;
}

__device__ __forceinline__ float wave_max64(float v) {
#pragma unroll
  for (int mask = 32; mask > 0; mask >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, mask, 64));
  return v;
}
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int mask = 32; mask > 0; mask >>= 1) v = v + __shfl_xor(v, mask, 64);
  return v;
}

}  // namespace omni
