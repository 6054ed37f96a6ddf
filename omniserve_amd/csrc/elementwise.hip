// Per-token activation quantisation, (general) RMS norm and SiLU*mul for MI355X (gfx950).
//
// Replaces omniserve_backend.fused_kernels / layernorm_ops / activation_ops
// (reference: kernels/csrc/fused_kernels.cu, layernorm_kernels.cu, activation_kernels.cu).
//
// The reference's results depend on its reduction geometry (1024 virtual threads per token, thread t
// accumulating elements t, t+1024, ... sequentially, then a 32-lane butterfly and a butterfly over 32 warp
// partials; the fuse_sum norm even accumulates per-thread in fp16).  The kernels in use (the "v2" row kernels
// below) keep that VIRTUAL geometry but run it on 128 / 256 / 512 physical threads per token: element values are
// computed in parallel, parked as f32 in LDS, and the first NV/8 threads replay the reference's per-virtual-thread
// accumulation order and reduction tree from there, so sums, scales and int8 codes are bit-identical to
// oracle/elementwise.py whatever the physical geometry.  The row is read from HBM exactly once instead of the
// reference's 3-4 passes.  (The first-generation kernels with one 1024-thread workgroup per token remain as the
// fallback for row lengths the v2 geometry does not cover.)
#include "common.h"
#include "row_reduce.h"
#include "row_kernels.h"
#include "tp_comm.h"
#include <cstdlib>

namespace omni {

constexpr int NT_MAX = 1024;
constexpr int VPT = 16;  // elements per thread held in registers: hidden <= 16384 fast path

// ------------------------------------------------------------------------------------------
// invoke_quant / invoke_quant_fuse_sum   (fused_kernels.cu:57-142)
// ------------------------------------------------------------------------------------------
// silu(x) = x / (1 + exp(-x)) in f32, rounded to fp16, times up in f32, rounded to fp16
// (activation_kernels.cu:10-13,84-97).  The reference is built with --use_fast_math, i.e. ex2.approx and an
// approximate division, so its f32 intermediate is not bit-defined; v_exp_f32 / v_rcp_f32 are the same class
// of approximation (<= 1 fp16 ulp from the exact value after rounding; tests allow 2).
// (silu_mul_h lives in common.h: the gate_up GEMV's fused epilogue uses the same function)

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
template <bool FUSE_SUM>
__global__ __launch_bounds__(NT_MAX) void silu_mul_quant_kernel(int8_t* __restrict__ out,
                                                                 const half_t* __restrict__ in,
                                                                 half_t* __restrict__ sum_out,
                                                                 half_t* __restrict__ scale_out, int d) {
  __shared__ float red[32];
  quant_row<FUSE_SUM>(out + (size_t)blockIdx.x * d, SiluMulLoader{in + (size_t)blockIdx.x * 2 * d, d}, sum_out,
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
    for (int j = 0; j < 8; ++j) o[j] = silu_mul_h(a[j], b[j]);
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
    out[t * d + c] = silu_mul_h(in[t * 2 * d + c], in[t * 2 * d + d + c]);
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
// A source is split into fetch() (loads only) and finish() (arithmetic, in-place side effects).  Sources
// with BATCH = true have their fetches for all of a thread's vectors issued back to back, branch-free
// (out-of-range vectors re-read element 0), before any finish(): these kernels are latency chains and a
// branch between two loads costs a whole extra memory round trip.
struct SrcPlain {
  static constexpr bool BATCH = true;
  struct Raw { v8h t; };
  const half_t* row;
  int stride;
  __device__ __forceinline__ void pin() const { asm volatile("" ::"s"(row), "s"(stride)); }   // see OMNI_PIN_ARGS
  __device__ __forceinline__ SrcPlain at_row(int m) const { return SrcPlain{row + (size_t)m * stride, stride}; }
  __device__ __forceinline__ void fetch(int i, Raw& r) const { r.t = *reinterpret_cast<const v8h*>(row + i); }
  __device__ __forceinline__ void finish(int i, const Raw& r, float (&x)[VT]) const {
#pragma unroll
    for (int e = 0; e < VT; ++e) x[e] = (float)r.t[e];
  }
};
struct SrcAdd {  // residual += delta (fp16 add), in place
  static constexpr bool BATCH = true;
  struct Raw { v8h a, d; };
  half_t* res;
  const half_t* delta;
  int stride;
  __device__ __forceinline__ void pin() const { asm volatile("" ::"s"(res), "s"(delta), "s"(stride)); }
  __device__ __forceinline__ SrcAdd at_row(int m) const {
    return SrcAdd{res + (size_t)m * stride, delta + (size_t)m * stride, stride};
  }
  __device__ __forceinline__ void fetch(int i, Raw& r) const {
    r.a = *reinterpret_cast<const v8h*>(res + i);
    r.d = *reinterpret_cast<const v8h*>(delta + i);
  }
  __device__ __forceinline__ void finish(int i, const Raw& r, float (&x)[VT]) const {
    v8h o;
#pragma unroll
    for (int e = 0; e < VT; ++e) { o[e] = (half_t)((float)r.a[e] + (float)r.d[e]); x[e] = (float)o[e]; }
    *reinterpret_cast<v8h*>(res + i) = o;
  }
};
// (SrcSlabAddT -- residual += h(GEMM epilogue(sum of split-K slabs)) -- lives in row_kernels.h: the fused MLP launch uses it too)
// tensor parallel: residual += h( sum over ranks of the peers' fp16 partial projections ) -- the all-reduce of
// llama_w4a8_unpad.py's row-parallel outputs folded into the consumer (tp_comm.h)
struct SrcPeerAdd {
  static constexpr bool BATCH = true;     // (all vectors' peer requests in flight together: one peer round trip per row)
  struct Raw { v8h a; v8h t[TP_MAX_WORLD]; };
  half_t* res;
  TpPeers tp;
  int stride;
  __device__ __forceinline__ void pin() const {}
  __device__ __forceinline__ SrcPeerAdd at_row(int m) const {
    SrcPeerAdd r = *this;
    r.res = res + (size_t)m * stride;
    r.tp.slot_off = tp.slot_off + (long long)m * stride;
    return r;
  }
  __device__ __forceinline__ void fetch(int i, Raw& r) const {
    r.a = *reinterpret_cast<const v8h*>(res + i);
    tp_fetch8(tp, (size_t)i, r.t);          // (a timed-out epoch still reads the slots -- mapped memory -- and drops them)
  }
  __device__ __forceinline__ void finish(int i, const Raw& r, float (&x)[VT]) const {
    const v8h s = tp_reduce8(tp, r.t, tp.epoch);
    v8h o;
#pragma unroll
    for (int e = 0; e < VT; ++e) { o[e] = (half_t)((float)r.a[e] + (float)s[e]); x[e] = (float)o[e]; }
    *reinterpret_cast<v8h*>(res + i) = o;
  }
};
struct SrcSilu {  // h(h(silu(gate)) * up) of a [2d] row
  static constexpr bool BATCH = true;
  struct Raw { v8h a, b; };
  const half_t* row;
  int d;
  __device__ __forceinline__ void pin() const { asm volatile("" ::"s"(row), "s"(d)); }
  __device__ __forceinline__ SrcSilu at_row(int m) const { return SrcSilu{row + (size_t)m * 2 * d, d}; }
  __device__ __forceinline__ void fetch(int i, Raw& r) const {
    r.a = *reinterpret_cast<const v8h*>(row + i);
    r.b = *reinterpret_cast<const v8h*>(row + d + i);
  }
  __device__ __forceinline__ void finish(int i, const Raw& r, float (&x)[VT]) const {
#pragma unroll
    for (int e = 0; e < VT; ++e) x[e] = (float)silu_mul_h(r.a[e], r.b[e]);
  }
};

// (SrcAttnMerge -- the flash-decoding merge of one token's split partials -- lives in row_kernels.h: the decode attention
// kernel's last-arriver merge uses the same code)

// Row kernels, final form: RT threads compute the element values in parallel (8 consecutive elements
// = one 16-B access per thread and iteration) and park them as f32 in LDS; the first NV/8 threads then
// replay the reference's per-virtual-thread accumulation ORDER from LDS (cheap: hidden/NV adds per
// value) and run the reference reduction tree.  Heavy per-element work (silu, split-K slab sums) is
// spread over the whole workgroup, the ordered part stays bit-identical to oracle/elementwise.py.
// Geometry <RT threads per row, RV 8-element vectors per thread>: <512, 4> for decode-size batches (the row is a
// latency chain: spread it over 8 waves), <128, 4> / <256, 8> when there are many rows (prefill: 4x / 2x more rows
// resident per CU, the kernel becomes bandwidth-bound instead of chain-bound).  hidden <= RT * RV * 8.

// quant[_fuse_sum] (NV = min(hidden,1024)) with a pluggable source
template <int RT, int RV, bool FUSE_SUM, typename Src>
__global__ __launch_bounds__(RT) void quant_v2_kernel(int8_t* __restrict__ out, Src src0,
                                                       half_t* __restrict__ sum_out, half_t* __restrict__ scale_out,
                                                       int hidden, int nv, PrefetchArgs pf) {
  extern __shared__ __attribute__((aligned(16))) float xs[];   // [hidden] (>= RT/64 KiB, see OMNI_V2_LAUNCH)
  __shared__ float red[96];
  // (the rider test reads two kernel arguments: the row path's own are requested in the same batch of scalar loads -- a test
  //  on its own made every row workgroup pay a second, dependent kernarg round trip in front of its first load)
  src0.pin();
  asm volatile("" ::"s"(out), "s"(sum_out), "s"(scale_out), "s"(hidden), "s"(nv), "s"(pf.blocks), "s"(pf.first_block));
  if (pf.blocks > 0 && (int)blockIdx.x >= pf.first_block) {    // extra workgroups: L2 prefetch for the next GEMV
    prefetch_weights_to_l2(pf, xs);
    return;
  }
  const int p = threadIdx.x;
  OMNI_CLK(8);
  const Src src = src0.at_row(blockIdx.x);
  float x[RV][VT];
  float amax = 0.0f;
  typename Src::Raw raw[Src::BATCH ? RV : 1];
  if constexpr (Src::BATCH) {
#pragma unroll
    for (int it = 0; it < RV; ++it) {
      const int i = (p + it * RT) * VT;
      src.fetch(i < hidden ? i : 0, raw[it]);
    }
  }
  OMNI_CLK(9);
#pragma unroll
  for (int it = 0; it < RV; ++it) {
    const int i = (p + it * RT) * VT;
    const bool ok = i < hidden;
    if (ok) {
      if constexpr (Src::BATCH) {
        src.finish(i, raw[it], x[it]);
      } else {
        src.fetch(i, raw[0]);
        src.finish(i, raw[0], x[it]);
      }
    }
#pragma unroll
    for (int e = 0; e < VT; ++e) {
      x[it][e] = ok ? x[it][e] : 0.0f;
      amax = __builtin_fmaxf(amax, __builtin_fabsf(x[it][e]));
    }
    if (FUSE_SUM && ok) {
      *reinterpret_cast<v4f*>(xs + i) = (v4f){x[it][0], x[it][1], x[it][2], x[it][3]};
      *reinterpret_cast<v4f*>(xs + i + 4) = (v4f){x[it][4], x[it][5], x[it][6], x[it][7]};
    }
  }
  OMNI_CLK(10);
  amax = block_max_rt<RT>(amax, red);   // (its barriers also publish xs)
  OMNI_CLK(11);
  // the codes only need the maximum: their stores go out first and drain while the ordered sum is replayed
  if (p == 0) scale_out[blockIdx.x] = (half_t)(amax / 127.0f);
  const float q = 127.0f / amax;
  int8_t* orow = out + (size_t)blockIdx.x * hidden;
#pragma unroll
  for (int it = 0; it < RV; ++it) {
    const int i = (p + it * RT) * VT;
    if (i < hidden) store8_i8(orow + i, x[it], q);
  }
  OMNI_CLK(12);
  if constexpr (FUSE_SUM) {
    float s[1][VT], tot[1];
    ordered_partials<1>(xs, p, nv, hidden, s, [](float (&v)[1][VT], int e, float val) { v[0][e] = v[0][e] + val; });
    tree_sum8<1>(s, red, p, nv >> 5, tot);
    if (p == 0) sum_out[blockIdx.x] = (half_t)tot[0];
  }
  OMNI_CLK(13);
}

// (general_norm_v2_body -- rms_norm_general[_fuse_sum] with a pluggable source and sink -- lives in row_kernels.h)
template <int RT, int RV, bool FUSE_SUM, typename Src>
__global__ __launch_bounds__(RT) void general_norm_v2_kernel(int8_t* __restrict__ out, Src src0, const half_t* __restrict__ gamma,
                                                              half_t* __restrict__ sum_out, half_t* __restrict__ scale_out,
                                                              float eps, int hidden, int nv, PrefetchArgs pf) {
  extern __shared__ __attribute__((aligned(16))) float xs[];   // [hidden] (>= RT/64 KiB, see OMNI_V2_LAUNCH)
  __shared__ float red[96];
  src0.pin();      // (as in quant_v2_kernel: one batch of scalar loads for the rider test and the row path)
  asm volatile("" ::"s"(out), "s"(gamma), "s"(sum_out), "s"(scale_out), "s"(eps), "s"(hidden), "s"(nv), "s"(pf.blocks), "s"(pf.first_block));
  if (pf.blocks > 0 && (int)blockIdx.x >= pf.first_block) {    // extra workgroups: L2 prefetch for the next GEMV
    prefetch_weights_to_l2(pf, xs);
    return;
  }
  general_norm_v2_body<RT, RV, FUSE_SUM, Src>(out, src0, gamma, sum_out, scale_out, eps, hidden, nv, xs, red);
}

// tensor parallel: the same row kernel behind the peers' barrier (tp_comm.h); one workgroup per token, all resident
template <int RT, int RV, bool FUSE_SUM>
__global__ __launch_bounds__(RT) void tp_add_norm_v2_kernel(int8_t* __restrict__ out, SrcPeerAdd src0, const half_t* __restrict__ gamma,
                                                             half_t* __restrict__ sum_out, half_t* __restrict__ scale_out,
                                                             float eps, int hidden, int nv) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  __shared__ float red[96];
  uint32_t e = tp_publish_and_wait(src0.tp);
  SinkGlobal sink{out + (size_t)blockIdx.x * hidden, sum_out, scale_out, (int)blockIdx.x};
  if (src0.tp.two_shot) {
    // two-shot form (tp_comm.h): chunk = whole rows; shot 1: the kernel's workgroups reduce THIS rank's rows into its gather
    // region; shot 2: every row adds its reduced projection from the row's owner -- residual += delta, SrcAdd's arithmetic,
    // the very expression SrcPeerAdd::finish applies to the sum: bit-identical to the one-shot form
    const TpPeers& tp = src0.tp;
    const long long count = (long long)gridDim.x * hidden;
    const long long c0 = (long long)tp.rank * tp.chunk;
    const long long mine_n = c0 >= count ? 0 : (count - c0 < tp.chunk ? count - c0 : tp.chunk);
    half_t* gather = const_cast<half_t*>(tp_data_of(tp, tp.rank)) + tp.gather_off;
    for (long long v = (long long)blockIdx.x * RT + threadIdx.x; v < mine_n / 8; v += (long long)gridDim.x * RT)
      *reinterpret_cast<v8h*>(gather + v * 8) = tp_sum8(tp, (size_t)(c0 + v * 8), e);
    if (tp_between_shots(tp, e)) {
      const long long i0 = (long long)blockIdx.x * hidden;
      const int owner = (int)(i0 / tp.chunk);
      const SrcAdd row{src0.res + i0, tp_data_of(tp, owner) + tp.gather_off + (i0 - (long long)owner * tp.chunk), hidden};
      general_norm_v2_row<RT, RV, FUSE_SUM, SrcAdd, SinkGlobal>(row, gamma, sink, eps, hidden, nv, xs, red);
      tp_finish(tp, e);
      return;
    }
    e = 0;       // a wait timed out: the row is poisoned below exactly as in the one-shot form
  }
  SrcPeerAdd src = src0.at_row(blockIdx.x);     // (a local copy: writing the epoch into the kernel argument parked it in scratch)
  src.tp.epoch = e;         // 0 = a peer never arrived: the row is poisoned (NaN), the epoch stays
  general_norm_v2_row<RT, RV, FUSE_SUM, SrcPeerAdd, SinkGlobal>(src, gamma, sink, eps, hidden, nv, xs, red);
  tp_finish(src0.tp, e);
}

// rms_norm (fp16 out): NV = min(hidden,1024)
// Src: SrcPlain (the reference's rms_norm) or a split-K slab consumer (fused extension: the LAST layer's down projection
// deferred into the model's final norm, omni_splitk_add_rms_norm)
template <int RT, int RV, typename Src = SrcPlain>
__global__ __launch_bounds__(RT) void rms_norm_v2_kernel(half_t* __restrict__ out, Src src0,
                                                          const half_t* __restrict__ weight, float eps, int hidden, int nv) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  __shared__ float red[96];
  const int p = threadIdx.x;
  const Src src = src0.at_row(blockIdx.x);
  float x[RV][VT];
  typename Src::Raw raw[Src::BATCH ? RV : 1];
  v8h w8[RV];
#pragma unroll
  for (int it = 0; it < RV; ++it) {
    const int i = (p + it * RT) * VT;
    const int ic = i < hidden ? i : 0;
    if constexpr (Src::BATCH) src.fetch(ic, raw[it]);
    w8[it] = *reinterpret_cast<const v8h*>(weight + ic);
  }
#pragma unroll
  for (int it = 0; it < RV; ++it) {
    const int i = (p + it * RT) * VT;
    if (i < hidden) {
      if constexpr (Src::BATCH) {
        src.finish(i, raw[it], x[it]);
      } else {
        src.fetch(i, raw[0]);
        src.finish(i, raw[0], x[it]);
      }
      *reinterpret_cast<v4f*>(xs + i) = (v4f){x[it][0], x[it][1], x[it][2], x[it][3]};
      *reinterpret_cast<v4f*>(xs + i + 4) = (v4f){x[it][4], x[it][5], x[it][6], x[it][7]};
    }
  }
  __syncthreads();
  float st[1][VT], tv[1];
  ordered_partials<1>(xs, p, nv, hidden, st, [](float (&v)[1][VT], int e, float val) { v[0][e] = v[0][e] + val * val; });
  tree_sum8<1>(st, red, p, nv >> 5, tv);
  const float rstd = 1.0f / __builtin_sqrtf(tv[0] / (float)hidden + eps);
  half_t* orow = out + (size_t)blockIdx.x * hidden;
#pragma unroll
  for (int it = 0; it < RV; ++it) {
    const int i = (p + it * RT) * VT;
    if (i < hidden) {
      v8h o;
#pragma unroll
      for (int e = 0; e < VT; ++e) {
        const half_t t = (half_t)rounded_f32(x[it][e] * rstd);
        o[e] = (half_t)((float)t * (float)w8[it][e]);
      }
      *reinterpret_cast<v8h*>(orow + i) = o;
    }
  }
}

// dispatch helpers ------------------------------------------------------------------------------------
static inline bool v2_ok(int hidden, int nv) {
  return hidden % 8 == 0 && nv % 32 == 0 && nv <= 1024 && hidden <= 512 * 4 * VT && (size_t)hidden * 4 <= 64 * 1024;
}
#ifndef OMNI_ROWS_MANY
#define OMNI_ROWS_MANY 1024
#endif
constexpr int ROWS_MANY = OMNI_ROWS_MANY;   // from here on the narrow geometries win (measured at 16384 rows)
// KERNEL(RT, RV) must expand to a kernel instantiation; `hidden` here is the row length that sizes the f32 LDS copy
// One-shot prefetch descriptor (omni_prefetch_arm_gemm, qgemm_plan.hip): consumed by the next v2 quant /
// general-norm launch.
PrefetchArgs take_armed_prefetch();
static inline PrefetchArgs take_prefetch(int tokens) {
  PrefetchArgs pf = take_armed_prefetch();
  pf.first_block = tokens;
  if (tokens >= ROWS_MANY) pf.blocks = 0;    // many-row launches are bandwidth-bound themselves: nothing idles
  return pf;
}
// (kernels without a PrefetchArgs parameter)
#define OMNI_V2_LAUNCH_PLAIN(KERNEL, tokens, hidden, elems, ...)                                                \
  do {                                                                                                          \
    const size_t lds_ = (size_t)(hidden) * sizeof(float);                                                       \
    if ((tokens) >= ROWS_MANY && (elems) <= 128 * 4 * VT)                                                       \
      hipLaunchKernelGGL((KERNEL(128, 4)), dim3(tokens), dim3(128), lds_, (hipStream_t)stream, __VA_ARGS__);    \
    else if ((tokens) >= ROWS_MANY && (elems) <= 256 * 8 * VT)                                                  \
      hipLaunchKernelGGL((KERNEL(256, 8)), dim3(tokens), dim3(256), lds_, (hipStream_t)stream, __VA_ARGS__);    \
    else                                                                                                        \
      hipLaunchKernelGGL((KERNEL(512, 4)), dim3(tokens), dim3(512), lds_, (hipStream_t)stream, __VA_ARGS__);    \
  } while (0)
// quant / general-norm kernels: the armed prefetch descriptor rides along as extra workgroups (decode-size launches).
// OMNI_DECODE_RT=256 (tuning knob) runs decode-size rows of sources that batch their loads on 256 threads instead of 512.
static const bool v2_batched = true;     // shadowed by `false` at the launch sites of sources that fetch per vector
static inline int decode_rt() {
  static const int v = omni_knob("OMNI_DECODE_RT", 512);
  return v;
}
#define OMNI_V2_LAUNCH(KERNEL, tokens, hidden, elems, ...)                                                      \
  do {                                                                                                          \
    const PrefetchArgs pf_ = take_prefetch(tokens);                                                             \
    size_t lds_ = (size_t)(hidden) * sizeof(float);                                                             \
    if (pf_.blocks > 0 && lds_ < 8 * 1024) lds_ = 8 * 1024;   /* 1 KiB of LDS-DMA target per wave */            \
    if ((tokens) >= ROWS_MANY && (elems) <= 128 * 4 * VT)                                                       \
      hipLaunchKernelGGL((KERNEL(128, 4)), dim3(tokens), dim3(128), lds_, (hipStream_t)stream, __VA_ARGS__, pf_);  \
    else if (((tokens) >= ROWS_MANY || (decode_rt() == 256 && v2_batched)) && (elems) <= 256 * 8 * VT)          \
      hipLaunchKernelGGL((KERNEL(256, 8)), dim3((tokens) + pf_.blocks), dim3(256), lds_, (hipStream_t)stream,   \
                         __VA_ARGS__, pf_);                                                                     \
    else                                                                                                        \
      hipLaunchKernelGGL((KERNEL(512, 4)), dim3((tokens) + pf_.blocks), dim3(512), lds_, (hipStream_t)stream,   \
                         __VA_ARGS__, pf_);                                                                     \
  } while (0)

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
    #undef KQ_
    #define KQ_(RT_, RV_) quant_v2_kernel<RT_, RV_, false, SrcPlain>
    OMNI_V2_LAUNCH(KQ_, tokens, 8, hidden, (int8_t*)out_i8, SrcPlain{(const half_t*)in_f16, hidden},
                   (half_t*)nullptr, (half_t*)scale_f16, hidden, nv);
    return omni_launch_status();
  }
  (void)take_armed_prefetch();   // this geometry cannot carry the L2 prefetch: drop an armed descriptor instead of leaving it for a later launch
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
    #undef KQ_
    #define KQ_(RT_, RV_) quant_v2_kernel<RT_, RV_, true, SrcPlain>
    OMNI_V2_LAUNCH(KQ_, tokens, hidden, hidden, (int8_t*)out_i8, SrcPlain{(const half_t*)in_f16, hidden},
                   (half_t*)sum_f16, (half_t*)scale_f16, hidden, nv);
    return omni_launch_status();
  }
  (void)take_armed_prefetch();   // this geometry cannot carry the L2 prefetch: drop an armed descriptor instead of leaving it for a later launch
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
      #undef KQ_
      #define KQ_(RT_, RV_) rms_norm_v2_kernel<RT_, RV_>
      OMNI_V2_LAUNCH_PLAIN(KQ_, tokens, hidden, hidden, (half_t*)out_f16, SrcPlain{(const half_t*)in_f16, hidden},
                           (const half_t*)weight_f16, eps, hidden, nv);
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
      #undef KQ_
      #define KQ_(RT_, RV_) general_norm_v2_kernel<RT_, RV_, false, SrcPlain>
      OMNI_V2_LAUNCH(KQ_, tokens, hidden, hidden, (int8_t*)out_i8,
                     SrcPlain{(const half_t*)in_f16, hidden}, (const half_t*)weight_f16, (half_t*)nullptr,
                     (half_t*)scale_f16, eps, hidden, nv);
      return omni_launch_status();
    }
  }
  (void)take_armed_prefetch();   // this geometry cannot carry the L2 prefetch: drop an armed descriptor instead of leaving it for a later launch
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
      #undef KQ_
      #define KQ_(RT_, RV_) general_norm_v2_kernel<RT_, RV_, true, SrcPlain>
      OMNI_V2_LAUNCH(KQ_, tokens, hidden, hidden, (int8_t*)out_i8,
                     SrcPlain{(const half_t*)in_f16, hidden}, (const half_t*)weight_f16, (half_t*)sum_f16,
                     (half_t*)scale_f16, eps, hidden, nv);
      return omni_launch_status();
    }
  }
  (void)take_armed_prefetch();   // this geometry cannot carry the L2 prefetch: drop an armed descriptor instead of leaving it for a later launch
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
  // sum_f16 == NULL: the row sum is not wanted (W8A8 / per-group callers: rms_norm_general without _fuse_sum upstream)
  if (!out_i8 || !residual_f16 || !delta_f16 || !weight_f16 || !scale_f16 || tokens < 0 || hidden < 1)
    return OMNI_EINVAL;
  if (hidden > VPT * NT_MAX) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  {
    const int nv = norm_block(hidden, true);
    if (v2_ok(hidden, nv)) {
      if (sum_f16) {
        #undef KQ_
        #define KQ_(RT_, RV_) general_norm_v2_kernel<RT_, RV_, true, SrcAdd>
        OMNI_V2_LAUNCH(KQ_, tokens, hidden, hidden, (int8_t*)out_i8,
                       SrcAdd{(half_t*)residual_f16, (const half_t*)delta_f16, hidden}, (const half_t*)weight_f16,
                       (half_t*)sum_f16, (half_t*)scale_f16, eps, hidden, nv);
      } else {
        #undef KQ_
        #define KQ_(RT_, RV_) general_norm_v2_kernel<RT_, RV_, false, SrcAdd>
        OMNI_V2_LAUNCH(KQ_, tokens, hidden, hidden, (int8_t*)out_i8,
                       SrcAdd{(half_t*)residual_f16, (const half_t*)delta_f16, hidden}, (const half_t*)weight_f16,
                       (half_t*)nullptr, (half_t*)scale_f16, eps, hidden, nv);
      }
      return omni_launch_status();
    }
  }
  (void)take_armed_prefetch();
  if (sum_f16)
    hipLaunchKernelGGL((general_norm_quant_kernel<true, true>), dim3(tokens), dim3(norm_block(hidden, true)),
                       0, (hipStream_t)stream, (int8_t*)out_i8, (half_t*)residual_f16, (const half_t*)delta_f16,
                       (const half_t*)weight_f16, (half_t*)sum_f16, (half_t*)scale_f16, eps, hidden);
  else
    hipLaunchKernelGGL((general_norm_quant_kernel<false, true>), dim3(tokens), dim3(norm_block(hidden, true)),
                       0, (hipStream_t)stream, (int8_t*)out_i8, (half_t*)residual_f16, (const half_t*)delta_f16,
                       (const half_t*)weight_f16, (half_t*)nullptr, (half_t*)scale_f16, eps, hidden);
  return omni_launch_status();
}

extern "C" int omni_silu_mul_quant_fuse_sum(void* out_i8, const void* in_f16, void* sum_f16, void* scale_f16,
                                            int tokens, int d, void* stream) {
  // sum_f16 == NULL: silu_and_mul + invoke_quant (no row sum: the W8A8 / per-group MLP, activation.py:66-82)
  if (!out_i8 || !in_f16 || !scale_f16 || tokens < 0 || d < 1) return OMNI_EINVAL;
  if (d % 32 != 0) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  {
    const int nv = norm_block(d, false);
    if (v2_ok(d, nv)) {
      if (sum_f16) {
        #undef KQ_
        #define KQ_(RT_, RV_) quant_v2_kernel<RT_, RV_, true, SrcSilu>
        OMNI_V2_LAUNCH(KQ_, tokens, d, d, (int8_t*)out_i8, SrcSilu{(const half_t*)in_f16, d},
                       (half_t*)sum_f16, (half_t*)scale_f16, d, nv);
      } else {
        #undef KQ_
        #define KQ_(RT_, RV_) quant_v2_kernel<RT_, RV_, false, SrcSilu>
        OMNI_V2_LAUNCH(KQ_, tokens, d, d, (int8_t*)out_i8, SrcSilu{(const half_t*)in_f16, d},
                       (half_t*)nullptr, (half_t*)scale_f16, d, nv);
      }
      return omni_launch_status();
    }
  }
  // Rows of 16384 < d <= 32768 (Llama-2/3-70B's intermediate size at TP = 1: 28672): the <512, 8> geometry, 112 KiB of LDS for
  // the f32 row copy (one workgroup per CU; a decode batch has fewer rows than the chip has CUs).  The one-thread-per-32-columns
  // kernel below took 13.6 us per layer at bs = 128 (128 workgroups, a serial loop of 56 vectors per thread; profiles/r05_f).
  {
    const int nv = norm_block(d, false);
    // decode-size launches only (measured there; a many-row launch would run ONE 512-thread workgroup per CU at 112 - 128 KiB
    // of LDS: those keep the kernel below -- ADVICE r5)
    if (tokens < ROWS_MANY && d > 512 * 4 * VT && d <= 512 * 8 * VT && d % 8 == 0 && nv % 32 == 0 && nv <= 1024) {
      const PrefetchArgs pf = take_prefetch(tokens);
      const size_t lds = (size_t)d * sizeof(float);
      bool raised_ok = true;
      auto launch = [&](auto kern, half_t* sum) {
        // (dynamic LDS beyond 64 KiB is an opt-in per kernel AND per device: set on every call, like sparse_utils.hip does --
        //  a process-wide flag missed the second device and raced between threads)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) != hipSuccess) {
          raised_ok = false;
          return;
        }
        hipLaunchKernelGGL(kern, dim3(tokens + pf.blocks), dim3(512), lds, (hipStream_t)stream, (int8_t*)out_i8,
                           SrcSilu{(const half_t*)in_f16, d}, sum, (half_t*)scale_f16, d, nv, pf);
      };
      if (sum_f16) launch(quant_v2_kernel<512, 8, true, SrcSilu>, (half_t*)sum_f16);
      else launch(quant_v2_kernel<512, 8, false, SrcSilu>, (half_t*)nullptr);
      return raised_ok ? omni_launch_status() : OMNI_ELAUNCH;
    }
  }
  (void)take_armed_prefetch();
  // row lengths outside every v2 geometry
  if (sum_f16)
    hipLaunchKernelGGL(silu_mul_quant_kernel<true>, dim3(tokens), dim3(norm_block(d, false)), 0, (hipStream_t)stream,
                       (int8_t*)out_i8, (const half_t*)in_f16, (half_t*)sum_f16, (half_t*)scale_f16, d);
  else
    hipLaunchKernelGGL(silu_mul_quant_kernel<false>, dim3(tokens), dim3(norm_block(d, false)), 0, (hipStream_t)stream,
                       (int8_t*)out_i8, (const half_t*)in_f16, (half_t*)nullptr, (half_t*)scale_f16, d);
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
                    (const half_t*)a_ssums_in_f16, (half_t)0.0f, (half_t)0.0f};
  if (tokens < ROWS_MANY && hidden > 512 * VT && hidden <= 512 * 2 * VT) {
    // decode-size rows of 4097 .. 8192 columns (Llama-2-70B): two vectors per thread with BATCHED requests -- the unbatched
    // source made the second vector's slab loads a second memory round trip of the row chain
    typedef SrcSlabAddT<true, true> SrcB;
    SrcB srcb{src.res, src.slab, src.sstride, src.sk, src.stride, src.wscales, src.wsz, src.ascales, src.asum, (half_t)0.0f, (half_t)0.0f};
    const PrefetchArgs pf = take_prefetch(tokens);
    size_t lds = (size_t)hidden * sizeof(float);
    if (pf.blocks > 0 && lds < 8 * 1024) lds = 8 * 1024;
    hipLaunchKernelGGL((general_norm_v2_kernel<512, 2, true, SrcB>), dim3(tokens + pf.blocks), dim3(512), lds, (hipStream_t)stream,
                       (int8_t*)out_i8, srcb, (const half_t*)weight_f16, (half_t*)sum_f16, (half_t*)scale_f16, eps, hidden, nv, pf);
    return omni_launch_status();
  }
  const bool v2_batched = false;
  #undef KQ_
  #define KQ_(RT_, RV_) general_norm_v2_kernel<RT_, RV_, true, SrcSlabAddChn>
  OMNI_V2_LAUNCH(KQ_, tokens, hidden, hidden, (int8_t*)out_i8, src,
                 (const half_t*)weight_f16, (half_t*)sum_f16, (half_t*)scale_f16, eps, hidden, nv);
  return omni_launch_status();
}

// The same for the W8A8 GEMM (omni_w8a8_gemm_partial): epilogue h(f32(acc) * (wscales[n] * ascales[m])), no zero term.
extern "C" int omni_splitk_w8_add_rms_norm_general_fuse_sum(void* out_i8, void* residual_f16, const void* slab_i32, int sk,
                                                            const void* wscales_f16, const void* ascales_in_f16,
                                                            const void* weight_f16, void* sum_f16, void* scale_f16,
                                                            float eps, int tokens, int hidden, void* stream) {
  // sum_f16 == NULL: no row sum (the W8A8 / per-group layers never read one)
  if (!out_i8 || !residual_f16 || !slab_i32 || !wscales_f16 || !ascales_in_f16 || !weight_f16 ||
      !scale_f16 || tokens < 0 || hidden < 1 || sk < 1)
    return OMNI_EINVAL;
  const int nv = norm_block(hidden, true);
  if (!v2_ok(hidden, nv)) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  SrcSlabAddW8 src{(half_t*)residual_f16, (const int32_t*)slab_i32, (size_t)tokens * hidden, sk, hidden,
                   (const half_t*)wscales_f16, (const half_t*)nullptr, (const half_t*)ascales_in_f16,
                   (const half_t*)nullptr, (half_t)0.0f, (half_t)0.0f};
  const bool v2_batched = false;
  if (sum_f16) {
    #undef KQ_
    #define KQ_(RT_, RV_) general_norm_v2_kernel<RT_, RV_, true, SrcSlabAddW8>
    OMNI_V2_LAUNCH(KQ_, tokens, hidden, hidden, (int8_t*)out_i8, src,
                   (const half_t*)weight_f16, (half_t*)sum_f16, (half_t*)scale_f16, eps, hidden, nv);
  } else {
    #undef KQ_
    #define KQ_(RT_, RV_) general_norm_v2_kernel<RT_, RV_, false, SrcSlabAddW8>
    OMNI_V2_LAUNCH(KQ_, tokens, hidden, hidden, (int8_t*)out_i8, src,
                   (const half_t*)weight_f16, (half_t*)nullptr, (half_t*)scale_f16, eps, hidden, nv);
  }
  return omni_launch_status();
}

// Fused extension: the LAST decoder layer's down projection deferred into the model's final norm -- residual += fp16(GEMM
// epilogue(sum of the split-K slabs)), then rms_norm (layernorm_kernels.cu rms_norm_kernel; llama_w4a8_unpad.py:484 self.norm).
// w_szs_f16 / a_ssums_in_f16 == NULL: the W8A8 / per-group epilogue acc * (sw * sa); else the per-channel one.
extern "C" int omni_splitk_add_rms_norm(void* out_f16, void* residual_f16, const void* slab_i32, int sk,
                                        const void* wscales_f16, const void* ascales_in_f16, const void* w_szs_f16,
                                        const void* a_ssums_in_f16, const void* weight_f16, float eps, int tokens, int hidden,
                                        void* stream) {
  if (!out_f16 || !residual_f16 || !slab_i32 || !wscales_f16 || !ascales_in_f16 || !weight_f16 || tokens < 0 || hidden < 1 ||
      sk < 1 || ((w_szs_f16 == nullptr) != (a_ssums_in_f16 == nullptr)))
    return OMNI_EINVAL;
  const int nv = norm_block(hidden, false);
  if (!v2_ok(hidden, nv)) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  if (w_szs_f16) {
    SrcSlabAddChn src{(half_t*)residual_f16, (const int32_t*)slab_i32, (size_t)tokens * hidden, sk, hidden,
                      (const half_t*)wscales_f16, (const half_t*)w_szs_f16, (const half_t*)ascales_in_f16,
                      (const half_t*)a_ssums_in_f16, (half_t)0.0f, (half_t)0.0f};
    #undef KQ_
    #define KQ_(RT_, RV_) rms_norm_v2_kernel<RT_, RV_, SrcSlabAddChn>
    OMNI_V2_LAUNCH_PLAIN(KQ_, tokens, hidden, hidden, (half_t*)out_f16, src, (const half_t*)weight_f16, eps, hidden, nv);
  } else {
    SrcSlabAddW8 src{(half_t*)residual_f16, (const int32_t*)slab_i32, (size_t)tokens * hidden, sk, hidden,
                     (const half_t*)wscales_f16, (const half_t*)nullptr, (const half_t*)ascales_in_f16,
                     (const half_t*)nullptr, (half_t)0.0f, (half_t)0.0f};
    #undef KQ_
    #define KQ_(RT_, RV_) rms_norm_v2_kernel<RT_, RV_, SrcSlabAddW8>
    OMNI_V2_LAUNCH_PLAIN(KQ_, tokens, hidden, hidden, (half_t*)out_f16, src, (const half_t*)weight_f16, eps, hidden, nv);
  }
  return omni_launch_status();
}

// Fused extension: kv4_decode_merge_kernel + omni_quant_fuse_sum in one kernel (one workgroup per token).
extern "C" int omni_attn_merge_quant_fuse_sum(void* out_i8, const void* part_ml_f32, const void* part_o_f32, int nsplit,
                                              void* sum_f16, void* scale_f16, int batch, int num_heads, void* stream) {
  // sum_f16 == NULL: merge + invoke_quant (no row sum)
  if (!out_i8 || !part_ml_f32 || !part_o_f32 || !scale_f16 || nsplit < 1 || batch < 0 || num_heads < 1)
    return OMNI_EINVAL;
  const int hidden = num_heads * 128;
  const int nv = norm_block(hidden, false);
  if (!v2_ok(hidden, nv)) return OMNI_EINVAL;
  if (batch == 0) return OMNI_OK;
  SrcAttnMerge src{(const float*)part_ml_f32, (const float*)part_o_f32, nsplit, num_heads, 0};
  const bool v2_batched = false;
  if (sum_f16) {
    #undef KQ_
    #define KQ_(RT_, RV_) quant_v2_kernel<RT_, RV_, true, SrcAttnMerge>
    OMNI_V2_LAUNCH(KQ_, batch, hidden, hidden, (int8_t*)out_i8, src, (half_t*)sum_f16,
                   (half_t*)scale_f16, hidden, nv);
  } else {
    #undef KQ_
    #define KQ_(RT_, RV_) quant_v2_kernel<RT_, RV_, false, SrcAttnMerge>
    OMNI_V2_LAUNCH(KQ_, batch, hidden, hidden, (int8_t*)out_i8, src, (half_t*)nullptr,
                   (half_t*)scale_f16, hidden, nv);
  }
  return omni_launch_status();
}

// Fused extension (row-kernel-free decode layer): the flash-decoding merge alone, as a WIDE kernel -- one thread per 8
// output elements, no block reduction -- writing the fp16 attention output (the values omni_kv4_decode_attention
// returns) and raising the row maxima of |out| in amax_slots (integer atomicMax on the f32 bits: exact, order
// independent).  The o projection then quantises on the fly (omni_w4a8_per_chn_gemm_partial_f16).  Carries the armed L2
// prefetch like the row kernels do.
__global__ __launch_bounds__(256) void attn_merge_f16_kernel(half_t* __restrict__ out, SrcAttnMerge src0,
                                                             uint32_t* __restrict__ amax, int batch, int hidden,
                                                             PrefetchArgs pf) {
  __shared__ __attribute__((aligned(16))) uint8_t dma_scratch[4 * 1024];
  if (pf.blocks > 0 && (int)blockIdx.x >= pf.first_block) {
    prefetch_weights_to_l2(pf, dma_scratch);
    return;
  }
  const int tpt = hidden / VT;                                  // threads per token (a multiple of 64)
  const int gt = blockIdx.x * 256 + threadIdx.x;
  const int token = gt / tpt;
  if (token >= batch) return;                                    // (whole waves: tpt % 64 == 0)
  const int i = (gt - token * tpt) * VT;
  const SrcAttnMerge src = src0.at_row(token);
  SrcAttnMerge::Raw raw;
  float x[VT];
  src.fetch(i, raw);
  src.finish(i, raw, x);
  v8h o;
  float mx = 0.0f;
#pragma unroll
  for (int e = 0; e < VT; ++e) {
    o[e] = (half_t)x[e];                                         // x[e] is already an fp16 value
    mx = __builtin_fmaxf(mx, __builtin_fabsf(x[e]));
  }
  *reinterpret_cast<v8h*>(out + (size_t)token * hidden + i) = o;
  mx = wave_max64(mx);
  if ((threadIdx.x & 63) == 0) amax_raise(amax, token, i >> 9, mx);
}

extern "C" int omni_attn_merge_f16_amax(void* out_f16, const void* part_ml_f32, const void* part_o_f32, int nsplit,
                                        void* amax_slots_u32, int batch, int num_heads, void* stream) {
  if (!out_f16 || !part_ml_f32 || !part_o_f32 || !amax_slots_u32 || nsplit < 1 || batch < 0 || num_heads < 1)
    return OMNI_EINVAL;
  if (num_heads % 4 != 0 || batch > AMAX_ROWS) return OMNI_EINVAL;   // a wave (512 elements) must not straddle two tokens
  if (batch == 0) return OMNI_OK;
  const int hidden = num_heads * 128;
  PrefetchArgs pf = take_armed_prefetch();
  const int wgs = (batch * (hidden / VT) + 255) / 256;
  pf.first_block = wgs;
  SrcAttnMerge src{(const float*)part_ml_f32, (const float*)part_o_f32, nsplit, num_heads, 0};
  hipLaunchKernelGGL(attn_merge_f16_kernel, dim3(wgs + pf.blocks), dim3(256), 0, (hipStream_t)stream, (half_t*)out_f16,
                     src, (uint32_t*)amax_slots_u32, batch, hidden, pf);
  return omni_launch_status();
}

// Tensor parallel (fused extension, tp_comm.h): residual += all-reduce(partial projections); rms_norm_general[_fuse_sum].
// The partial projection of THIS rank must already sit in its slot (written by the projection's launch).
extern "C" int omni_tp_add_rms_norm_general_fuse_sum(void* out_i8, void* residual_f16, const void* const* peer_data,
                                                     void* const* peer_flags, int rank, int world,
                                                     long long slot_offset_elems, const void* weight_f16, void* sum_f16,
                                                     void* scale_f16, float eps, int tokens, int hidden,
                                                     long long gather_offset_elems, int algo, void* stream) {
  if (!out_i8 || !residual_f16 || !weight_f16 || !scale_f16 || !peer_data || !peer_flags || tokens < 0 || hidden < 1)
    return OMNI_EINVAL;
  if (algo < 0 || algo > 2 || (algo == 2 && gather_offset_elems < 0) || hidden % 8 != 0) return OMNI_EINVAL;
  if (world < 1 || world > TP_MAX_WORLD || rank < 0 || rank >= world || slot_offset_elems < 0) return OMNI_EINVAL;
  const int nv = norm_block(hidden, true);
  if (!v2_ok(hidden, nv) || tokens > 256) return OMNI_EINVAL;     // (every token's workgroup must be resident: they meet at a ticket)
  if (tokens == 0) return OMNI_OK;
  SrcPeerAdd src{};
  src.res = (half_t*)residual_f16; src.stride = hidden;
  for (int p = 0; p < TP_MAX_WORLD; ++p) {
    const int q = p < world ? p : 0;
    if (!peer_data[q] || !peer_flags[q]) return OMNI_EINVAL;
    src.tp.data[p] = (const half_t*)peer_data[q];
    src.tp.flags[p] = (uint32_t*)peer_flags[q];
  }
  src.tp.rank = rank; src.tp.world = world; src.tp.slot_off = slot_offset_elems; src.tp.epoch = 1;
  // two shots (tp_comm.h) from TP_TWO_SHOT_BYTES of payload on more than two ranks (algo 0), or forced (algo 2): chunk = whole rows
  src.tp.gather_off = 0; src.tp.chunk = 0; src.tp.two_shot = 0;
  if (gather_offset_elems >= 0 && algo != 1 && (algo == 2 || (world > 2 && (long long)tokens * hidden * 2 >= TP_TWO_SHOT_BYTES))) {
    src.tp.two_shot = 1;
    src.tp.gather_off = gather_offset_elems;
    src.tp.chunk = (long long)((tokens + world - 1) / world) * hidden;
  }
  // this row kernel carries no L2-prefetch riders (its workgroups meet at a ticket and must all be resident): a descriptor
  // armed for "the next row kernel" is dropped here instead of riding on an unrelated launch later
  (void)take_armed_prefetch();
  const size_t lds = (size_t)hidden * sizeof(float);
  // vectors per thread = what the row needs (the source batches its requests: an unused vector would re-read element 0 of
  // every peer): 1 up to 4096 columns, 2 up to 8192 (Llama-2-70B), 4 beyond
#define OMNI_TP_NORM(RV_)                                                                                                         \
  do {                                                                                                                            \
    if (sum_f16)                                                                                                                  \
      hipLaunchKernelGGL((tp_add_norm_v2_kernel<512, RV_, true>), dim3(tokens), dim3(512), lds, (hipStream_t)stream,              \
                         (int8_t*)out_i8, src, (const half_t*)weight_f16, (half_t*)sum_f16, (half_t*)scale_f16, eps, hidden, nv); \
    else                                                                                                                          \
      hipLaunchKernelGGL((tp_add_norm_v2_kernel<512, RV_, false>), dim3(tokens), dim3(512), lds, (hipStream_t)stream,             \
                         (int8_t*)out_i8, src, (const half_t*)weight_f16, (half_t*)nullptr, (half_t*)scale_f16, eps, hidden, nv); \
  } while (0)
  if (hidden <= 512 * VT) OMNI_TP_NORM(1);
  else if (hidden <= 512 * 2 * VT) OMNI_TP_NORM(2);
  else OMNI_TP_NORM(4);
#undef OMNI_TP_NORM
  return omni_launch_status();
}

// ------------------------------------------------------------------------------------------
// Greedy sampling helper for the decode runner: out[r] = index of the first maximum of logits[r, :]
// (torch.argmax semantics: NaN counts as the maximum, -0 == +0).  Two tiny kernels, no state:
// (chunks, rows) workgroups reduce 64-bit keys (orderable fp16 bits << 32 | ~index), one workgroup per row
// finishes.  torch's generic reduce kernel takes 47 us for [16, 128256] inside the decode step; this takes ~6.
// ------------------------------------------------------------------------------------------
constexpr int ARGMAX_CHUNKS = 32;

__device__ __forceinline__ unsigned long long argmax_key(uint32_t h16, uint32_t idx) {
  uint32_t h = h16 & 0xFFFFu;
  if (h == 0x8000u) h = 0;                                   // -0 == +0
  const uint32_t k = (h & 0x8000u) ? (~h & 0xFFFFu) : (h | 0x8000u);
  return ((unsigned long long)k << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)v, m, 64), hi = __shfl_xor((uint32_t)(v >> 32), m, 64);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    v = o > v ? o : v;
  }
  return v;
}

__global__ __launch_bounds__(256) void argmax_partial_kernel(const half_t* __restrict__ x, int64_t row_stride, int cols,
                                                             unsigned long long* __restrict__ part) {
  __shared__ unsigned long long red[4];
  const int row = blockIdx.y, chunk = blockIdx.x;
  const int per = ((cols + ARGMAX_CHUNKS - 1) / ARGMAX_CHUNKS + 7) & ~7;
  const int c0 = chunk * per, c1 = min(cols, c0 + per);
  const half_t* xr = x + (size_t)row * row_stride;
  unsigned long long best = 0;
  const bool vec = (row_stride % 8) == 0 && (reinterpret_cast<uintptr_t>(x) % 16) == 0;
  if (vec) {
    for (int c = c0 + 8 * threadIdx.x; c < c1; c += 8 * 256) {
      if (c + 8 <= c1) {
        const uint4 v = *reinterpret_cast<const uint4*>(xr + c);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned long long k0 = argmax_key(w[j], c + 2 * j), k1 = argmax_key(w[j] >> 16, c + 2 * j + 1);
          best = k0 > best ? k0 : best;
          best = k1 > best ? k1 : best;
        }
      } else {
        for (int e = c; e < c1; ++e) {
          const unsigned long long k = argmax_key(__builtin_bit_cast(uint16_t, xr[e]), e);
          best = k > best ? k : best;
        }
      }
    }
  } else {
    for (int e = c0 + threadIdx.x; e < c1; e += 256) {
      const unsigned long long k = argmax_key(__builtin_bit_cast(uint16_t, xr[e]), e);
      best = k > best ? k : best;
    }
  }
  best = wave_max_u64(best);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long b = red[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) b = red[w] > b ? red[w] : b;
    part[(size_t)row * ARGMAX_CHUNKS + chunk] = b;
  }
}

__global__ __launch_bounds__(64) void argmax_final_kernel(const unsigned long long* __restrict__ part,
                                                          long long* __restrict__ out) {
  const int row = blockIdx.x;
  unsigned long long v = threadIdx.x < ARGMAX_CHUNKS ? part[(size_t)row * ARGMAX_CHUNKS + threadIdx.x] : 0ull;
  v = wave_max_u64(v);
  if (threadIdx.x == 0) out[row] = (long long)(0xFFFFFFFFu - (uint32_t)(v & 0xFFFFFFFFull));
}

// Embedding lookup of the decode drivers (torch.nn.Embedding upstream, llama_w4a8_unpad.py:480): out[r, :] = table[idx[r], :].
// torch.index_select runs a 12.6-us kernel for 16 rows; this one is a plain row copy (16 B per lane).
__global__ __launch_bounds__(256) void gather_rows_kernel(half_t* __restrict__ out, const half_t* __restrict__ table,
                                                           const int64_t* __restrict__ idx, int cols, int64_t table_rows) {
  const int64_t r = idx[blockIdx.x];
  uint4* dst = reinterpret_cast<uint4*>(out + (size_t)blockIdx.x * cols);
  if (r < 0 || r >= table_rows) {
    // torch.index_select raises on an out-of-range id; a kernel inside a captured graph cannot: the row is filled
    // with NaN so the sequence visibly fails instead of decoding from the previous step's stale embedding
    for (int i = threadIdx.x; i < cols / 8; i += 256) dst[i] = make_uint4(0x7E007E00u, 0x7E007E00u, 0x7E007E00u, 0x7E007E00u);
    return;
  }
  const uint4* src = reinterpret_cast<const uint4*>(table + (size_t)r * cols);
  for (int i = threadIdx.x; i < cols / 8; i += 256) dst[i] = src[i];
}

// Decode drivers, first launch of a step: the embedding lookup + lengths[b] += 1 + the zeroing of the step's row-maximum
// slots in ONE launch (three dependent ~4.7-us launches otherwise: lengths.add_(1), amax.zero_(), the lookup).
__global__ __launch_bounds__(256) void decode_step_begin_kernel(half_t* __restrict__ out, const half_t* __restrict__ table,
                                                                const int64_t* __restrict__ idx, int cols, int64_t table_rows,
                                                                int* __restrict__ lengths, int n_lengths,
                                                                uint32_t* __restrict__ zero, long long zero_words) {
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < n_lengths; i += 256) lengths[i] += 1;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < zero_words; i += (long long)gridDim.x * 256) zero[i] = 0u;
  const int64_t r = idx[blockIdx.x];
  uint4* dst = reinterpret_cast<uint4*>(out + (size_t)blockIdx.x * cols);
  if (r < 0 || r >= table_rows) {     // (see gather_rows_kernel)
    for (int i = threadIdx.x; i < cols / 8; i += 256) dst[i] = make_uint4(0x7E007E00u, 0x7E007E00u, 0x7E007E00u, 0x7E007E00u);
    return;
  }
  const uint4* src = reinterpret_cast<const uint4*>(table + (size_t)r * cols);
  for (int i = threadIdx.x; i < cols / 8; i += 256) dst[i] = src[i];
}

extern "C" int omni_decode_step_begin(void* out_f16, const void* table_f16, const void* idx_i64, int rows, int cols,
                                      int64_t table_rows, void* lengths_i32, int n_lengths, void* zero_u32,
                                      long long zero_words, void* stream) {
  if (!out_f16 || !table_f16 || !idx_i64 || rows < 1 || cols < 8 || cols % 8 || table_rows < 1 || n_lengths < 0 ||
      zero_words < 0 || (n_lengths > 0 && !lengths_i32) || (zero_words > 0 && !zero_u32))
    return OMNI_EINVAL;
  hipLaunchKernelGGL(decode_step_begin_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (half_t*)out_f16,
                     (const half_t*)table_f16, (const int64_t*)idx_i64, cols, table_rows, (int*)lengths_i32, n_lengths,
                     (uint32_t*)zero_u32, zero_words);
  return omni_launch_status();
}

extern "C" int omni_gather_rows_f16(void* out_f16, const void* table_f16, const void* idx_i64, int rows, int cols,
                                    int64_t table_rows, void* stream) {
  if (!out_f16 || !table_f16 || !idx_i64 || rows < 0 || cols < 8 || cols % 8 || table_rows < 1) return OMNI_EINVAL;
  if (rows == 0) return OMNI_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (half_t*)out_f16,
                     (const half_t*)table_f16, (const int64_t*)idx_i64, cols, table_rows);
  return omni_launch_status();
}

extern "C" size_t omni_argmax_workspace_bytes(int rows) {
  return rows > 0 ? (size_t)rows * ARGMAX_CHUNKS * sizeof(unsigned long long) : 0;
}

extern "C" int omni_argmax_f16(void* out_i64, const void* logits_f16, int64_t row_stride, int rows, int cols,
                               void* workspace, size_t workspace_bytes, void* stream) {
  if (!out_i64 || !logits_f16 || !workspace || rows < 0 || cols < 1 || row_stride < cols) return OMNI_EINVAL;
  if (workspace_bytes < omni_argmax_workspace_bytes(rows)) return OMNI_ENOMEM;
  if (rows == 0) return OMNI_OK;
  hipLaunchKernelGGL(argmax_partial_kernel, dim3(ARGMAX_CHUNKS, rows), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)logits_f16, row_stride, cols, (unsigned long long*)workspace);
  hipLaunchKernelGGL(argmax_final_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream,
                     (const unsigned long long*)workspace, (long long*)out_i64);
  return omni_launch_status();
}

OMNI_CLK_READER(omni_debug_clocks_elementwise)
