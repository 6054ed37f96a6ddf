// Per-token activation quantisation, (general) RMS norm and SiLU*mul for MI355X (gfx950).
//
// Replaces omniserve_backend.fused_kernels / layernorm_ops / activation_ops
// (reference: kernels/csrc/fused_kernels.cu, layernorm_kernels.cu, activation_kernels.cu).
//
// The reference's results depend on its reduction geometry (1024 virtual threads per token,
// thread t accumulating elements t, t+1024, ... sequentially, then a 32-lane butterfly and a
// butterfly over 32 warp partials; the fuse_sum norm even accumulates per-thread in fp16).
// These kernels keep that geometry -- one 1024-thread workgroup per token, virtual warp =
// 32-lane half of a wave64 -- so sums, scales and int8 codes are bit-identical to
// oracle/elementwise.py.  Each thread keeps its (<= VPT) elements in registers, so the row is
// read from HBM exactly once instead of the reference's 3-4 passes.
#include "common.h"

namespace omni {

constexpr int NT_MAX = 1024;
constexpr int VPT = 16;  // elements per thread held in registers: hidden <= 16384 fast path

// ------------------------------------------------------------------------------------------
// invoke_quant / invoke_quant_fuse_sum   (fused_kernels.cu:57-142)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ half_t silu_mul_h(half_t a, half_t b) {
  const float xf = (float)a;
  const half_t s = (half_t)(xf / (1.0f + expf(-xf)));
  return (half_t)((float)s * (float)b);
}

struct PlainLoader {   // x[i] of a contiguous fp16 row
  const half_t* row;
  __device__ __forceinline__ float operator()(int i) const { return (float)row[i]; }
};
struct SiluMulLoader { // h(h(silu(gate[i])) * up[i]) of a [2d] row: the value silu_and_mul would store
  const half_t* row;
  int d;
  __device__ __forceinline__ float operator()(int i) const { return (float)silu_mul_h(row[i], row[d + i]); }
};

template <bool FUSE_SUM, typename Loader>
__device__ __forceinline__ void quant_row(int8_t* __restrict__ out_row, Loader ld, half_t* __restrict__ sum_out,
                                          half_t* __restrict__ scale_out, int hidden, float* red) {
  const int tid = threadIdx.x, nt = blockDim.x;
  float x[VPT];
  float amax = 0.0f, s = 0.0f;
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int i = tid + j * nt;
    const bool ok = i < hidden;  // branch-free: keeps x[] in registers
    const float v = ld(ok ? i : 0);
    x[j] = ok ? v : 0.0f;
    if constexpr (FUSE_SUM) s = ok ? s + x[j] : s;
    amax = __builtin_fmaxf(amax, __builtin_fabsf(x[j]));
  }
  for (int i = tid + VPT * nt; i < hidden; i += nt) {  // hidden > VPT*1024: re-evaluated below
    const float v = ld(i);
    if constexpr (FUSE_SUM) s = s + v;
    amax = __builtin_fmaxf(amax, __builtin_fabsf(v));
  }
  amax = ref_block_max(amax, red, -1e20f);
  if constexpr (FUSE_SUM) {
    const float tot = ref_block_sum(s, red);
    if (tid == 0) sum_out[blockIdx.x] = (half_t)tot;
  }
  if (tid == 0) scale_out[blockIdx.x] = (half_t)(amax / 127.0f);
  const float q = 127.0f / amax;
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int i = tid + j * nt;
    if (i < hidden) out_row[i] = rni_sat_s8(x[j] * q);
  }
  for (int i = tid + VPT * nt; i < hidden; i += nt) out_row[i] = rni_sat_s8(ld(i) * q);
}

template <bool FUSE_SUM>
__global__ __launch_bounds__(NT_MAX) void quant_kernel(int8_t* __restrict__ out,
                                                        const half_t* __restrict__ in,
                                                        half_t* __restrict__ sum_out,
                                                        half_t* __restrict__ scale_out, int hidden) {
  __shared__ float red[32];
  const size_t row = (size_t)blockIdx.x * hidden;
  quant_row<FUSE_SUM>(out + row, PlainLoader{in + row}, sum_out, scale_out, hidden, red);
}

// Fused extension: silu_and_mul + invoke_quant_fuse_sum without the fp16
// [tokens, d] round trip; bit-identical to running the two kernels back to back.
__global__ __launch_bounds__(NT_MAX) void silu_mul_quant_kernel(int8_t* __restrict__ out,
                                                                 const half_t* __restrict__ in,
                                                                 half_t* __restrict__ sum_out,
                                                                 half_t* __restrict__ scale_out, int d) {
  __shared__ float red[32];
  quant_row<true>(out + (size_t)blockIdx.x * d, SiluMulLoader{in + (size_t)blockIdx.x * 2 * d, d}, sum_out,
                  scale_out, d, red);
}

// ------------------------------------------------------------------------------------------
// rms_norm_general[_fuse_sum], per-token quant   (layernorm_kernels.cu:58-331)
//   y = (x - mean) * rsqrt(mean(x^2) + eps) * gamma   [mean subtracted in the output only]
// ------------------------------------------------------------------------------------------
// ADD: fused extension -- `in` is the residual stream, updated in place with
// x = h(x + delta) (the torch fp16 add the reference does between the two calls) before the norm.
template <bool FUSE_SUM, bool ADD>
__global__ __launch_bounds__(NT_MAX) void general_norm_quant_kernel(
    int8_t* __restrict__ out, half_t* __restrict__ in, const half_t* __restrict__ delta,
    const half_t* __restrict__ gamma, half_t* __restrict__ sum_out, half_t* __restrict__ scale_out,
    float eps, int hidden) {
  __shared__ float red[32];
  const int tid = threadIdx.x, nt = blockDim.x;
  const size_t row = (size_t)blockIdx.x * hidden;
  float x[VPT];
  float lsum = 0.0f, lsq = 0.0f;
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int i = tid + j * nt;
    const bool ok = i < hidden;
    float v = (float)in[row + (ok ? i : 0)];
    if constexpr (ADD) {
      const half_t xs = (half_t)(v + (float)delta[row + (ok ? i : 0)]);
      if (ok) in[row + i] = xs;
      v = (float)xs;
    }
    x[j] = ok ? v : 0.0f;
    lsum = ok ? lsum + x[j] : lsum;
    lsq = ok ? lsq + x[j] * x[j] : lsq;
  }
  const float mean = ref_block_sum(lsum, red) / (float)hidden;
  const float var = ref_block_sum(lsq, red);
  const float rstd = 1.0f / __builtin_sqrtf(var / (float)hidden + eps);

  // amax / sum are fp16 quantities in the reference; fp16 values are exact in f32, so they are
  // carried as floats and re-rounded to fp16 where the reference rounds.
  float amax_h = (float)(half_t)1e-6f;
  float hsum = 0.0f;
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int i = tid + j * nt;
    const bool ok = i < hidden;
    float y = (x[j] - mean) * rstd;
    y = rounded_f32(y * (float)gamma[ok ? i : 0]);
    x[j] = y;
    const float yh = ok ? (float)(half_t)y : 0.0f;
    amax_h = __builtin_fmaxf(amax_h, __builtin_fabsf(yh));
    if constexpr (FUSE_SUM) hsum = ok ? (float)(half_t)(hsum + yh) : hsum;
  }
  const float amax = ref_block_max(amax_h, red, -1e20f);
  if constexpr (FUSE_SUM) {
    const float tot = ref_block_sum(hsum, red);
    if (tid == 0) sum_out[blockIdx.x] = (half_t)tot;
  }
  if (tid == 0) scale_out[blockIdx.x] = (half_t)(amax / 127.0f);
  const float q = 127.0f / amax;
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int i = tid + j * nt;
    if (i < hidden) out[row + i] = rni_sat_s8(x[j] * q);
  }
}

// ------------------------------------------------------------------------------------------
// rms_norm (fp16 out)   (layernorm_kernels.cu:335-365)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT_MAX) void rms_norm_kernel(half_t* __restrict__ out,
                                                           const half_t* __restrict__ in,
                                                           const half_t* __restrict__ weight,
                                                           float eps, int hidden) {
  __shared__ float red[32];
  const int tid = threadIdx.x, nt = blockDim.x;
  const size_t row = (size_t)blockIdx.x * hidden;
  float x[VPT];
  float lsq = 0.0f;
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int i = tid + j * nt;
    const bool ok = i < hidden;
    const float v = (float)in[row + (ok ? i : 0)];
    x[j] = ok ? v : 0.0f;
    lsq = ok ? lsq + x[j] * x[j] : lsq;
  }
  const float var = ref_block_sum(lsq, red);
  const float rstd = 1.0f / __builtin_sqrtf(var / (float)hidden + eps);
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int i = tid + j * nt;
    if (i < hidden) {
      const half_t t = (half_t)rounded_f32(x[j] * rstd);
      out[row + i] = (half_t)((float)t * (float)weight[i]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// silu_and_mul   (activation_kernels.cu:10-30): out = h( f32(h(x/(1+exp(-x)))) * f32(y) )
// 8 elements (16 B) per lane, grid-stride over tokens*d/8.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void silu_and_mul_kernel(half_t* __restrict__ out,
                                                            const half_t* __restrict__ in,
                                                            int tokens, int d) {
  const int vec_per_row = d / 8;
  const size_t total = (size_t)tokens * vec_per_row;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const size_t t = idx / vec_per_row;
    const int c = (int)(idx % vec_per_row) * 8;
    const v8h a = *reinterpret_cast<const v8h*>(in + t * 2 * d + c);
    const v8h b = *reinterpret_cast<const v8h*>(in + t * 2 * d + d + c);
    v8h o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xf = (float)a[j];
      const half_t s = (half_t)(xf / (1.0f + expf(-xf)));
      o[j] = (half_t)((float)s * (float)b[j]);
    }
    *reinterpret_cast<v8h*>(out + t * d + c) = o;
  }
}

__global__ __launch_bounds__(256) void silu_and_mul_scalar_kernel(half_t* __restrict__ out,
                                                                   const half_t* __restrict__ in,
                                                                   int tokens, int d) {
  const size_t total = (size_t)tokens * d;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const size_t t = idx / d;
    const int c = (int)(idx % d);
    const float xf = (float)in[t * 2 * d + c];
    const half_t s = (half_t)(xf / (1.0f + expf(-xf)));
    out[t * d + c] = (half_t)((float)s * (float)in[t * 2 * d + d + c]);
  }
}

// ==========================================================================================
// v2 row kernels: same arithmetic and the same (virtual) reduction geometry as above, executed by
// NV/8 physical threads.  Physical thread p owns the 8 consecutive virtual threads 8p..8p+7, i.e.
// 8 consecutive elements (one 16-B load) of every NV-wide chunk; a virtual warp (32 lanes) is 4
// adjacent physical threads.  The reference's butterfly (masks 16,8 across threads, 4,2,1 across
// a thread's 8 values) and its second-level butterfly over 32 warp partials are reproduced
// term for term, so results stay bit-identical to oracle/elementwise.py while a token needs
// 2 waves instead of 16 (fewer barrier hops, 16-B accesses): these kernels are pure latency chains.
// ==========================================================================================
constexpr int VT = 8;

// sum over all NV virtual threads of per-virtual-thread partials v[8] (two quantities at once)
template <int NQ>
__device__ __forceinline__ void tree_sum8(float (&v)[NQ][VT], float* red, int p, int nvwarps, float (&out)[NQ]) {
  float w[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
#pragma unroll
    for (int e = 0; e < VT; ++e) v[q][e] = v[q][e] + __shfl_xor(v[q][e], 2, 64);  // virtual mask 16
#pragma unroll
    for (int e = 0; e < VT; ++e) v[q][e] = v[q][e] + __shfl_xor(v[q][e], 1, 64);  // virtual mask 8
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

__device__ __forceinline__ float block_max_small(float m, float* red) {
  m = wave_max64(m);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[64 + (threadIdx.x >> 6)] = m;
  __syncthreads();
  float r = red[64];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = __builtin_fmaxf(r, red[64 + w]);
  return r;
}

// ---- sources: 8 consecutive fp16-valued elements starting at element i of a row --------------------
struct SrcPlain {
  const half_t* row;
  int stride;
  __device__ __forceinline__ SrcPlain at_row(int m) const { return SrcPlain{row + (size_t)m * stride, stride}; }
  __device__ __forceinline__ void load8(int i, float (&x)[VT]) const {
    const v8h t = *reinterpret_cast<const v8h*>(row + i);
#pragma unroll
    for (int e = 0; e < VT; ++e) x[e] = (float)t[e];
  }
};
struct SrcAdd {  // residual += delta (fp16 add), in place
  half_t* res;
  const half_t* delta;
  int stride;
  __device__ __forceinline__ SrcAdd at_row(int m) const {
    return SrcAdd{res + (size_t)m * stride, delta + (size_t)m * stride, stride};
  }
  __device__ __forceinline__ void load8(int i, float (&x)[VT]) const {
    const v8h a = *reinterpret_cast<const v8h*>(res + i);
    const v8h d = *reinterpret_cast<const v8h*>(delta + i);
    v8h o;
#pragma unroll
    for (int e = 0; e < VT; ++e) { o[e] = (half_t)((float)a[e] + (float)d[e]); x[e] = (float)o[e]; }
    *reinterpret_cast<v8h*>(res + i) = o;
  }
};
struct SrcSlabAddChn {  // residual += h(per-channel GEMM epilogue(sum of split-K slabs)), in place
  half_t* res;
  const int32_t* slab;    // [sk][M][N]
  size_t sstride;         // M*N
  int sk, stride;         // stride = N = hidden
  const half_t* wscales;  // [N]
  const half_t* wsz;      // [N]
  const half_t* ascales;  // [M] scales / sums of the GEMM's int8 input
  const half_t* asum;
  float sa, as;
  __device__ __forceinline__ SrcSlabAddChn at_row(int m) const {
    SrcSlabAddChn r = *this;
    r.res = res + (size_t)m * stride;
    r.slab = slab + (size_t)m * stride;
    r.sa = (float)ascales[m];
    r.as = (float)asum[m];
    return r;
  }
  __device__ __forceinline__ void load8(int i, float (&x)[VT]) const {
    v4i s0 = (v4i){0, 0, 0, 0}, s1 = s0;
    for (int k = 0; k < sk; ++k) {
      s0 += *reinterpret_cast<const v4i*>(slab + (size_t)k * sstride + i);
      s1 += *reinterpret_cast<const v4i*>(slab + (size_t)k * sstride + i + 4);
    }
    const v8h a = *reinterpret_cast<const v8h*>(res + i);
    const v8h sw = *reinterpret_cast<const v8h*>(wscales + i);
    const v8h sz = *reinterpret_cast<const v8h*>(wsz + i);
    v8h o;
#pragma unroll
    for (int e = 0; e < VT; ++e) {
      const int acc = e < 4 ? s0[e] : s1[e - 4];
      float t = (float)acc * (float)sw[e];
      t = t * sa;
      const float c = (float)sz[e] * as;
      const half_t ep = (half_t)(t - c);                       // = the GEMM's fp16 output
      o[e] = (half_t)((float)a[e] + (float)ep);
      x[e] = (float)o[e];
    }
    *reinterpret_cast<v8h*>(res + i) = o;
  }
};
struct SrcSilu {  // h(h(silu(gate)) * up) of a [2d] row
  const half_t* row;
  int d;
  __device__ __forceinline__ SrcSilu at_row(int m) const { return SrcSilu{row + (size_t)m * 2 * d, d}; }
  __device__ __forceinline__ void load8(int i, float (&x)[VT]) const {
    const v8h a = *reinterpret_cast<const v8h*>(row + i);
    const v8h b = *reinterpret_cast<const v8h*>(row + d + i);
#pragma unroll
    for (int e = 0; e < VT; ++e) x[e] = (float)silu_mul_h(a[e], b[e]);
  }
};

__device__ __forceinline__ void store8_i8(int8_t* dst, const float (&x)[VT], float q) {
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    lo |= (uint32_t)(uint8_t)rni_sat_s8(x[e] * q) << (8 * e);
    hi |= (uint32_t)(uint8_t)rni_sat_s8(x[4 + e] * q) << (8 * e);
  }
  *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
}

// quant[_fuse_sum]: NV = min(hidden, 1024)
template <int J, bool FUSE_SUM, typename Src>
__global__ __launch_bounds__(128) void quant_v2_kernel(int8_t* __restrict__ out, Src src0,
                                                        half_t* __restrict__ sum_out, half_t* __restrict__ scale_out,
                                                        int hidden, int nv) {
  __shared__ float red[96];
  const int p = threadIdx.x;
  const Src src = src0.at_row(blockIdx.x);
  float x[J][VT];
  float s[1][VT];
#pragma unroll
  for (int e = 0; e < VT; ++e) s[0][e] = 0.0f;
  float amax = 0.0f;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int i = j * nv + VT * p;
    const bool ok = VT * p < nv && i < hidden;
    if (ok) src.load8(i, x[j]);
#pragma unroll
    for (int e = 0; e < VT; ++e) {
      x[j][e] = ok ? x[j][e] : 0.0f;
      if constexpr (FUSE_SUM) s[0][e] = ok ? s[0][e] + x[j][e] : s[0][e];
      amax = __builtin_fmaxf(amax, __builtin_fabsf(x[j][e]));
    }
  }
  amax = block_max_small(amax, red);
  if constexpr (FUSE_SUM) {
    float tot[1];
    tree_sum8<1>(s, red, p, nv >> 5, tot);
    if (p == 0) sum_out[blockIdx.x] = (half_t)tot[0];
  }
  if (p == 0) scale_out[blockIdx.x] = (half_t)(amax / 127.0f);
  const float q = 127.0f / amax;
  int8_t* orow = out + (size_t)blockIdx.x * hidden;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int i = j * nv + VT * p;
    if (VT * p < nv && i < hidden) store8_i8(orow + i, x[j], q);
  }
}

// rms_norm_general[_fuse_sum] (+ fused residual sources): NV = roundup32(min(hidden,1024))
template <int J, bool FUSE_SUM, typename Src>
__global__ __launch_bounds__(128) void general_norm_v2_kernel(int8_t* __restrict__ out, Src src0, const half_t* __restrict__ gamma,
                                                               half_t* __restrict__ sum_out, half_t* __restrict__ scale_out,
                                                               float eps, int hidden, int nv) {
  __shared__ float red[96];
  const int p = threadIdx.x;
  const Src src = src0.at_row(blockIdx.x);
  float x[J][VT];
  float st[2][VT];
#pragma unroll
  for (int e = 0; e < VT; ++e) { st[0][e] = 0.0f; st[1][e] = 0.0f; }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int i = j * nv + VT * p;
    const bool ok = VT * p < nv && i < hidden;
    if (ok) src.load8(i, x[j]);
#pragma unroll
    for (int e = 0; e < VT; ++e) {
      x[j][e] = ok ? x[j][e] : 0.0f;
      st[0][e] = ok ? st[0][e] + x[j][e] : st[0][e];
      st[1][e] = ok ? st[1][e] + x[j][e] * x[j][e] : st[1][e];
    }
  }
  float tv[2];
  tree_sum8<2>(st, red, p, nv >> 5, tv);
  const float mean = tv[0] / (float)hidden;
  const float rstd = 1.0f / __builtin_sqrtf(tv[1] / (float)hidden + eps);
  float amax_h = (float)(half_t)1e-6f;
  float hs[1][VT];
#pragma unroll
  for (int e = 0; e < VT; ++e) hs[0][e] = 0.0f;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int i = j * nv + VT * p;
    const bool ok = VT * p < nv && i < hidden;
    v8h g8 = {};
    if (ok) g8 = *reinterpret_cast<const v8h*>(gamma + i);
#pragma unroll
    for (int e = 0; e < VT; ++e) {
      float y = (x[j][e] - mean) * rstd;
      y = rounded_f32(y * (float)g8[e]);
      x[j][e] = y;
      const float yh = ok ? (float)(half_t)y : 0.0f;
      amax_h = __builtin_fmaxf(amax_h, __builtin_fabsf(yh));
      if constexpr (FUSE_SUM) hs[0][e] = ok ? (float)(half_t)(hs[0][e] + yh) : hs[0][e];
    }
  }
  const float amax = block_max_small(amax_h, red);
  if constexpr (FUSE_SUM) {
    float tot[1];
    tree_sum8<1>(hs, red, p, nv >> 5, tot);
    if (p == 0) sum_out[blockIdx.x] = (half_t)tot[0];
  }
  if (p == 0) scale_out[blockIdx.x] = (half_t)(amax / 127.0f);
  const float q = 127.0f / amax;
  int8_t* orow = out + (size_t)blockIdx.x * hidden;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int i = j * nv + VT * p;
    if (VT * p < nv && i < hidden) store8_i8(orow + i, x[j], q);
  }
}

// rms_norm (fp16 out): NV = min(hidden,1024)
template <int J>
__global__ __launch_bounds__(128) void rms_norm_v2_kernel(half_t* __restrict__ out, const half_t* __restrict__ in,
                                                           const half_t* __restrict__ weight, float eps, int hidden, int nv) {
  __shared__ float red[96];
  const int p = threadIdx.x;
  const SrcPlain src{in + (size_t)blockIdx.x * hidden, hidden};
  float x[J][VT];
  float st[1][VT];
#pragma unroll
  for (int e = 0; e < VT; ++e) st[0][e] = 0.0f;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int i = j * nv + VT * p;
    const bool ok = VT * p < nv && i < hidden;
    if (ok) src.load8(i, x[j]);
#pragma unroll
    for (int e = 0; e < VT; ++e) {
      x[j][e] = ok ? x[j][e] : 0.0f;
      st[0][e] = ok ? st[0][e] + x[j][e] * x[j][e] : st[0][e];
    }
  }
  float tv[1];
  tree_sum8<1>(st, red, p, nv >> 5, tv);
  const float rstd = 1.0f / __builtin_sqrtf(tv[0] / (float)hidden + eps);
  half_t* orow = out + (size_t)blockIdx.x * hidden;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int i = j * nv + VT * p;
    if (VT * p < nv && i < hidden) {
      const v8h w8 = *reinterpret_cast<const v8h*>(weight + i);
      v8h o;
#pragma unroll
      for (int e = 0; e < VT; ++e) {
        const half_t t = (half_t)rounded_f32(x[j][e] * rstd);
        o[e] = (half_t)((float)t * (float)w8[e]);
      }
      *reinterpret_cast<v8h*>(orow + i) = o;
    }
  }
}

// dispatch helpers ------------------------------------------------------------------------------------
static inline bool v2_ok(int hidden, int nv) { return hidden % 8 == 0 && nv % 32 == 0 && hidden <= 16 * nv; }
static inline int v2_chunks(int hidden, int nv) {
  const int need = (hidden + nv - 1) / nv;
  return need <= 1 ? 1 : (need <= 2 ? 2 : (need <= 4 ? 4 : (need <= 8 ? 8 : 16)));
}
static inline dim3 v2_block(int nv) { return dim3(((nv / 8) + 63) / 64 * 64); }

#define OMNI_V2_DISPATCH_J(J_, CALL)            \
  switch (J_) {                                 \
    case 1: { constexpr int J = 1; CALL; } break;   \
    case 2: { constexpr int J = 2; CALL; } break;   \
    case 4: { constexpr int J = 4; CALL; } break;   \
    case 8: { constexpr int J = 8; CALL; } break;   \
    default: { constexpr int J = 16; CALL; } break; \
  }

static inline int norm_block(int hidden, bool round32) {
  int b = hidden < NT_MAX ? hidden : NT_MAX;
  if (round32) b = 32 * ((b + 31) / 32);
  return b;
}

}  // namespace omni

using namespace omni;

extern "C" int omni_quant(void* out_i8, const void* in_f16, void* scale_f16, int tokens, int hidden,
                          void* stream) {
  if (!out_i8 || !in_f16 || !scale_f16 || tokens < 0 || hidden < 1) return OMNI_EINVAL;
  if (hidden % 32 != 0) return OMNI_EINVAL;  // reference geometry: block = min(hidden,1024)
  if (tokens == 0) return OMNI_OK;
  const int nv = norm_block(hidden, false);
  if (v2_ok(hidden, nv)) {
    OMNI_V2_DISPATCH_J(v2_chunks(hidden, nv),
                       hipLaunchKernelGGL((quant_v2_kernel<J, false, SrcPlain>), dim3(tokens), v2_block(nv), 0,
                                          (hipStream_t)stream, (int8_t*)out_i8, SrcPlain{(const half_t*)in_f16, hidden},
                                          (half_t*)nullptr, (half_t*)scale_f16, hidden, nv));
    return omni_launch_status();
  }
  hipLaunchKernelGGL((quant_kernel<false>), dim3(tokens), dim3(norm_block(hidden, false)), 0,
                     (hipStream_t)stream, (int8_t*)out_i8, (const half_t*)in_f16, (half_t*)nullptr,
                     (half_t*)scale_f16, hidden);
  return omni_launch_status();
}

extern "C" int omni_quant_fuse_sum(void* out_i8, const void* in_f16, void* sum_f16, void* scale_f16,
                                   int tokens, int hidden, void* stream) {
  if (!out_i8 || !in_f16 || !sum_f16 || !scale_f16 || tokens < 0 || hidden < 1) return OMNI_EINVAL;
  if (hidden % 32 != 0) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  const int nv = norm_block(hidden, false);
  if (v2_ok(hidden, nv)) {
    OMNI_V2_DISPATCH_J(v2_chunks(hidden, nv),
                       hipLaunchKernelGGL((quant_v2_kernel<J, true, SrcPlain>), dim3(tokens), v2_block(nv), 0,
                                          (hipStream_t)stream, (int8_t*)out_i8, SrcPlain{(const half_t*)in_f16, hidden},
                                          (half_t*)sum_f16, (half_t*)scale_f16, hidden, nv));
    return omni_launch_status();
  }
  hipLaunchKernelGGL((quant_kernel<true>), dim3(tokens), dim3(norm_block(hidden, false)), 0,
                     (hipStream_t)stream, (int8_t*)out_i8, (const half_t*)in_f16, (half_t*)sum_f16,
                     (half_t*)scale_f16, hidden);
  return omni_launch_status();
}

extern "C" int omni_rms_norm(void* out_f16, const void* in_f16, const void* weight_f16, float eps,
                             int tokens, int hidden, void* stream) {
  if (!out_f16 || !in_f16 || !weight_f16 || tokens < 0 || hidden < 1) return OMNI_EINVAL;
  if (hidden % 32 != 0 || hidden > VPT * NT_MAX) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  {
    const int nv = norm_block(hidden, false);
    if (v2_ok(hidden, nv)) {
      OMNI_V2_DISPATCH_J(v2_chunks(hidden, nv),
                         hipLaunchKernelGGL((rms_norm_v2_kernel<J>), dim3(tokens), v2_block(nv), 0, (hipStream_t)stream,
                                            (half_t*)out_f16, (const half_t*)in_f16, (const half_t*)weight_f16, eps,
                                            hidden, nv));
      return omni_launch_status();
    }
  }
  hipLaunchKernelGGL(rms_norm_kernel, dim3(tokens), dim3(norm_block(hidden, false)), 0,
                     (hipStream_t)stream, (half_t*)out_f16, (const half_t*)in_f16,
                     (const half_t*)weight_f16, eps, hidden);
  return omni_launch_status();
}

extern "C" int omni_rms_norm_general(void* out_i8, const void* in_f16, const void* weight_f16,
                                     void* scale_f16, float eps, int tokens, int hidden,
                                     void* stream) {
  if (!out_i8 || !in_f16 || !weight_f16 || !scale_f16 || tokens < 0 || hidden < 1) return OMNI_EINVAL;
  if (hidden > VPT * NT_MAX) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  {
    const int nv = norm_block(hidden, true);
    if (v2_ok(hidden, nv)) {
      OMNI_V2_DISPATCH_J(v2_chunks(hidden, nv),
                         hipLaunchKernelGGL((general_norm_v2_kernel<J, false, SrcPlain>), dim3(tokens), v2_block(nv), 0,
                                            (hipStream_t)stream, (int8_t*)out_i8, SrcPlain{(const half_t*)in_f16, hidden},
                                            (const half_t*)weight_f16, (half_t*)nullptr, (half_t*)scale_f16, eps, hidden, nv));
      return omni_launch_status();
    }
  }
  hipLaunchKernelGGL((general_norm_quant_kernel<false, false>), dim3(tokens), dim3(norm_block(hidden, true)),
                     0, (hipStream_t)stream, (int8_t*)out_i8, (half_t*)in_f16, (const half_t*)nullptr,
                     (const half_t*)weight_f16, (half_t*)nullptr, (half_t*)scale_f16, eps, hidden);
  return omni_launch_status();
}

extern "C" int omni_rms_norm_general_fuse_sum(void* out_i8, const void* in_f16,
                                              const void* weight_f16, void* sum_f16,
                                              void* scale_f16, float eps, int tokens, int hidden,
                                              void* stream) {
  if (!out_i8 || !in_f16 || !weight_f16 || !sum_f16 || !scale_f16 || tokens < 0 || hidden < 1)
    return OMNI_EINVAL;
  if (hidden > VPT * NT_MAX) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  {
    const int nv = norm_block(hidden, true);
    if (v2_ok(hidden, nv)) {
      OMNI_V2_DISPATCH_J(v2_chunks(hidden, nv),
                         hipLaunchKernelGGL((general_norm_v2_kernel<J, true, SrcPlain>), dim3(tokens), v2_block(nv), 0,
                                            (hipStream_t)stream, (int8_t*)out_i8, SrcPlain{(const half_t*)in_f16, hidden},
                                            (const half_t*)weight_f16, (half_t*)sum_f16, (half_t*)scale_f16, eps, hidden, nv));
      return omni_launch_status();
    }
  }
  hipLaunchKernelGGL((general_norm_quant_kernel<true, false>), dim3(tokens), dim3(norm_block(hidden, true)),
                     0, (hipStream_t)stream, (int8_t*)out_i8, (half_t*)in_f16, (const half_t*)nullptr,
                     (const half_t*)weight_f16, (half_t*)sum_f16, (half_t*)scale_f16, eps, hidden);
  return omni_launch_status();
}

extern "C" int omni_silu_and_mul(void* out_f16, const void* in_f16, int tokens, int d, void* stream) {
  if (!out_f16 || !in_f16 || tokens < 0 || d < 1) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  const size_t total = (size_t)tokens * d;
  if (d % 8 == 0) {
    size_t blocks = (total / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(silu_and_mul_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (half_t*)out_f16, (const half_t*)in_f16, tokens, d);
  } else {
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(silu_and_mul_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, (half_t*)out_f16, (const half_t*)in_f16, tokens, d);
  }
  return omni_launch_status();
}

// ---- debug probe: the norm statistics exactly as general_norm_quant_kernel computes them ---------------
__global__ __launch_bounds__(NT_MAX) void norm_stats_debug_kernel(const half_t* __restrict__ in, float* __restrict__ out,
                                                                   float eps, int hidden) {
  __shared__ float red[32];
  const int tid = threadIdx.x, nt = blockDim.x;
  const size_t row = (size_t)blockIdx.x * hidden;
  float lsum = 0.0f, lsq = 0.0f;
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int i = tid + j * nt;
    const bool ok = i < hidden;
    const float v = ok ? (float)in[row + (ok ? i : 0)] : 0.0f;
    lsum = ok ? lsum + v : lsum;
    lsq = ok ? lsq + v * v : lsq;
  }
  const float tot = ref_block_sum(lsum, red);
  const float mean = tot / (float)hidden;
  const float var = ref_block_sum(lsq, red);
  const float vh = var / (float)hidden;
  const float ve = vh + eps;
  const float sq = __builtin_sqrtf(ve);
  const float rstd = 1.0f / sq;
  if (tid == 0) {
    float* o = out + (size_t)blockIdx.x * 8;
    o[0] = tot; o[1] = var; o[2] = mean; o[3] = vh; o[4] = ve; o[5] = sq; o[6] = rstd; o[7] = lsum;
  }
}

extern "C" int omni_debug_norm_stats(const void* in_f16, void* out_f32, float eps, int tokens, int hidden, void* stream) {
  hipLaunchKernelGGL(norm_stats_debug_kernel, dim3(tokens), dim3(norm_block(hidden, true)), 0, (hipStream_t)stream,
                     (const half_t*)in_f16, (float*)out_f32, eps, hidden);
  return omni_launch_status();
}

// ---- fused extensions (opt-in; not part of the reference API) ------------------------
extern "C" int omni_add_rms_norm_general_fuse_sum(void* out_i8, void* residual_f16, const void* delta_f16,
                                                  const void* weight_f16, void* sum_f16, void* scale_f16,
                                                  float eps, int tokens, int hidden, void* stream) {
  if (!out_i8 || !residual_f16 || !delta_f16 || !weight_f16 || !sum_f16 || !scale_f16 || tokens < 0 || hidden < 1)
    return OMNI_EINVAL;
  if (hidden > VPT * NT_MAX) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  {
    const int nv = norm_block(hidden, true);
    if (v2_ok(hidden, nv)) {
      OMNI_V2_DISPATCH_J(v2_chunks(hidden, nv),
                         hipLaunchKernelGGL((general_norm_v2_kernel<J, true, SrcAdd>), dim3(tokens), v2_block(nv), 0,
                                            (hipStream_t)stream, (int8_t*)out_i8,
                                            SrcAdd{(half_t*)residual_f16, (const half_t*)delta_f16, hidden},
                                            (const half_t*)weight_f16, (half_t*)sum_f16, (half_t*)scale_f16, eps, hidden, nv));
      return omni_launch_status();
    }
  }
  hipLaunchKernelGGL((general_norm_quant_kernel<true, true>), dim3(tokens), dim3(norm_block(hidden, true)),
                     0, (hipStream_t)stream, (int8_t*)out_i8, (half_t*)residual_f16, (const half_t*)delta_f16,
                     (const half_t*)weight_f16, (half_t*)sum_f16, (half_t*)scale_f16, eps, hidden);
  return omni_launch_status();
}

extern "C" int omni_silu_mul_quant_fuse_sum(void* out_i8, const void* in_f16, void* sum_f16, void* scale_f16,
                                            int tokens, int d, void* stream) {
  if (!out_i8 || !in_f16 || !sum_f16 || !scale_f16 || tokens < 0 || d < 1) return OMNI_EINVAL;
  if (d % 32 != 0) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  {
    const int nv = norm_block(d, false);
    if (v2_ok(d, nv)) {
      OMNI_V2_DISPATCH_J(v2_chunks(d, nv),
                         hipLaunchKernelGGL((quant_v2_kernel<J, true, SrcSilu>), dim3(tokens), v2_block(nv), 0,
                                            (hipStream_t)stream, (int8_t*)out_i8, SrcSilu{(const half_t*)in_f16, d},
                                            (half_t*)sum_f16, (half_t*)scale_f16, d, nv));
      return omni_launch_status();
    }
  }
  hipLaunchKernelGGL(silu_mul_quant_kernel, dim3(tokens), dim3(norm_block(d, false)), 0, (hipStream_t)stream,
                     (int8_t*)out_i8, (const half_t*)in_f16, (half_t*)sum_f16, (half_t*)scale_f16, d);
  return omni_launch_status();
}

extern "C" int omni_splitk_add_rms_norm_general_fuse_sum(void* out_i8, void* residual_f16, const void* slab_i32, int sk,
                                                         const void* wscales_f16, const void* ascales_in_f16,
                                                         const void* w_szs_f16, const void* a_ssums_in_f16,
                                                         const void* weight_f16, void* sum_f16, void* scale_f16,
                                                         float eps, int tokens, int hidden, void* stream) {
  if (!out_i8 || !residual_f16 || !slab_i32 || !wscales_f16 || !ascales_in_f16 || !w_szs_f16 || !a_ssums_in_f16 ||
      !weight_f16 || !sum_f16 || !scale_f16 || tokens < 0 || hidden < 1 || sk < 1)
    return OMNI_EINVAL;
  const int nv = norm_block(hidden, true);
  if (!v2_ok(hidden, nv)) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  SrcSlabAddChn src{(half_t*)residual_f16, (const int32_t*)slab_i32, (size_t)tokens * hidden, sk, hidden,
                    (const half_t*)wscales_f16, (const half_t*)w_szs_f16, (const half_t*)ascales_in_f16,
                    (const half_t*)a_ssums_in_f16, 0.f, 0.f};
  OMNI_V2_DISPATCH_J(v2_chunks(hidden, nv),
                     hipLaunchKernelGGL((general_norm_v2_kernel<J, true, SrcSlabAddChn>), dim3(tokens), v2_block(nv), 0,
                                        (hipStream_t)stream, (int8_t*)out_i8, src, (const half_t*)weight_f16,
                                        (half_t*)sum_f16, (half_t*)scale_f16, eps, hidden, nv));
  return omni_launch_status();
}
