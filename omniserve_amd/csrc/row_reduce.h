// Reference-order row reductions shared by the row kernels (elementwise.hip, offpath.hip).
//
// The reference reduces a row with 1024 (or min(hidden,1024)) threads: thread t accumulates elements t, t+NV, ...
// sequentially, then a 32-lane xor butterfly and a butterfly over the 32 warp partials
// (kernels/csrc/reduction_utils.cuh:25-164).  Here NV/8 physical threads replay that order: physical thread p owns
// the 8 consecutive virtual threads 8p..8p+7, i.e. 8 consecutive elements (one 16-B access) of every NV-wide chunk.
#pragma once
#include "common.h"

namespace omni {

constexpr int VT = 8;

// sum over all NV virtual threads of per-virtual-thread partials v[8] (two quantities at once)
template <int NQ>
__device__ __forceinline__ void tree_sum8(float (&v)[NQ][VT], float* red, int p, int nvwarps, float (&out)[NQ]) {
  float w[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
#pragma unroll
    for (int e = 0; e < VT; ++e) v[q][e] = v[q][e] + lane_xor2(v[q][e]);  // virtual mask 16 (same pairs as shfl_xor 2)
#pragma unroll
    for (int e = 0; e < VT; ++e) v[q][e] = v[q][e] + lane_xor1(v[q][e]);  // virtual mask 8
    const float c0 = v[q][0] + v[q][4], c1 = v[q][1] + v[q][5], c2 = v[q][2] + v[q][6], c3 = v[q][3] + v[q][7];
    const float d0 = c0 + c2, d1 = c1 + c3;
    w[q] = d0 + d1;
  }
  __syncthreads();
  if ((p & 3) == 0 && (p >> 2) < nvwarps) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) red[q * 32 + (p >> 2)] = w[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    float r[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) r[i] = i < nvwarps ? red[q * 32 + i] : 0.0f;
#pragma unroll
    for (int h = 16; h > 0; h >>= 1)
#pragma unroll
      for (int i = 0; i < h; ++i) r[i] = r[i] + r[i + h];
    out[q] = r[0];
  }
}

// the eight int8 codes rni_sat_s8(x[e] * q), packed little-endian
__device__ __forceinline__ uint2 pack8_i8(const float (&x)[VT], float q) {
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    lo |= (uint32_t)(uint8_t)rni_sat_s8(x[e] * q) << (8 * e);
    hi |= (uint32_t)(uint8_t)rni_sat_s8(x[4 + e] * q) << (8 * e);
  }
  return make_uint2(lo, hi);
}
__device__ __forceinline__ void store8_i8(int8_t* dst, const float (&x)[VT], float q) {
  *reinterpret_cast<uint2*>(dst) = pack8_i8(x, q);
}

template <int RT>
__device__ __forceinline__ float block_max_rt(float m, float* red) {
  m = wave_max64(m);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[64 + (threadIdx.x >> 6)] = m;
  __syncthreads();
  float r = red[64];
#pragma unroll
  for (int w = 1; w < RT / 64; ++w) r = __builtin_fmaxf(r, red[64 + w]);
  return r;
}

// ordered per-virtual-thread accumulation of NQ quantities from the LDS copy of the row
template <int NQ, typename F>
__device__ __forceinline__ void ordered_partials(const float* xs, int p, int nv, int hidden, float (&v)[NQ][VT], F f) {
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int e = 0; e < VT; ++e) v[q][e] = 0.0f;
  if (VT * p < nv) {
    // chunks in ascending order (that IS the reference's order); four chunks' LDS reads are issued together
    for (int i = VT * p; i < hidden; i += 4 * nv) {
      v4f a[4], b[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ic = (i + c * nv) < hidden ? i + c * nv : i;
        a[c] = *reinterpret_cast<const v4f*>(xs + ic);
        b[c] = *reinterpret_cast<const v4f*>(xs + ic + 4);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if ((i + c * nv) < hidden) {
#pragma unroll
          for (int e = 0; e < VT; ++e) f(v, e, e < 4 ? a[c][e] : b[c][e - 4]);
        }
      }
    }
  }
}

}  // namespace omni
