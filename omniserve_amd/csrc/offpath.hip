// The overloads of omniserve_backend.fused_kernels / layernorm_ops / activation_ops that the Llama W4A8 / W8A8 model
// code never calls but that belong to the modules' surface: static (per-tensor) scales, int32 -> fp16
// dequantisation, the fused "dequant + residual + RMS norm + quant" of the SmoothQuant-style W8A8 path, GELU.
// Reference: kernels/csrc/fused_kernels.cu:24-55,88-93,145-216,238-253; layernorm_kernels.cu:58-196,335-409,
// 432-468,515-561; activation_kernels.cu:31-131,186-213.  Oracle: oracle/elementwise.py (second half).
//
// All of them are HBM-bound row operations (2-8 bytes in, 1-2 bytes out per element):
//  * pointwise ones run one 16-B (fp16) or 2 x 16-B (int32) access per lane, a workgroup per 2048 elements of a row;
//  * the three norms keep the reference's VIRTUAL reduction geometry (min(hidden,1024) threads, thread t summing
//    elements t, t+NV, ... in order, then the 32-lane / 32-warp butterflies) on 256 physical threads through
//    row_reduce.h, so the statistics -- and with them the int8 codes -- are bit-identical to the oracle; the row is
//    read from HBM once;
//  * a * b + c of the reference is one FMA (nvcc's default contraction), written as such here.
#include "common.h"
#include "row_reduce.h"

namespace omni {

typedef int v8i __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void load8_i32(const int32_t* p, float (&x)[VT]) {
  const v4i a = *reinterpret_cast<const v4i*>(p), b = *reinterpret_cast<const v4i*>(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { x[e] = (float)a[e]; x[4 + e] = (float)b[e]; }
}
__device__ __forceinline__ void store8_codes(int8_t* dst, const float (&x)[VT]) {
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    lo |= (uint32_t)(uint8_t)rni_sat_s8(x[e]) << (8 * e);
    hi |= (uint32_t)(uint8_t)rni_sat_s8(x[4 + e]) << (8 * e);
  }
  *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
}

// ---------------------------------------------------------------------------------------------------------------
// pointwise kernels: grid (tokens, ceil(cols/2048)), 256 threads, 8 elements per thread
// ---------------------------------------------------------------------------------------------------------------
constexpr int PT = 256;

// invoke_quant / invoke_quant_fuse_sum, at::Half scale: q = rni_sat(x / scale)
__global__ __launch_bounds__(PT) void quant_static_kernel(int8_t* __restrict__ out, const half_t* __restrict__ in,
                                                           float scale, int hidden) {
  const int i = (blockIdx.y * PT + threadIdx.x) * VT;
  if (i >= hidden) return;
  const size_t o = (size_t)blockIdx.x * hidden + i;
  const v8h t = *reinterpret_cast<const v8h*>(in + o);
  float x[VT];
#pragma unroll
  for (int e = 0; e < VT; ++e) x[e] = (float)t[e] / scale;
  store8_codes(out + o, x);
}

// invoke_dequant: out = h(acc * scale), row strides in elements
__global__ __launch_bounds__(PT) void dequant_kernel(half_t* __restrict__ out, const int32_t* __restrict__ in,
                                                      float scale, int hidden, long long in_stride,
                                                      long long out_stride) {
  const int i = (blockIdx.y * PT + threadIdx.x) * VT;
  if (i >= hidden) return;
  float x[VT];
  load8_i32(in + (size_t)blockIdx.x * in_stride + i, x);
  v8h o;
#pragma unroll
  for (int e = 0; e < VT; ++e) o[e] = (half_t)rounded_f32(x[e] * scale);
  *reinterpret_cast<v8h*>(out + (size_t)blockIdx.x * out_stride + i) = o;
}

// invoke_dequant_add_residual: out = h(fma(acc, scale, residual)); per-token scale if `tok_scale`
__global__ __launch_bounds__(PT) void dequant_add_residual_kernel(half_t* __restrict__ out,
                                                                   const int32_t* __restrict__ in,
                                                                   const half_t* __restrict__ residual,
                                                                   const half_t* __restrict__ tok_scale, float scale,
                                                                   int hidden) {
  const int i = (blockIdx.y * PT + threadIdx.x) * VT;
  if (i >= hidden) return;
  const size_t o = (size_t)blockIdx.x * hidden + i;
  const float s = tok_scale ? (float)tok_scale[blockIdx.x] : scale;
  const v8h r = *reinterpret_cast<const v8h*>(residual + o);
  float x[VT];
  load8_i32(in + o, x);
  v8h y;
#pragma unroll
  for (int e = 0; e < VT; ++e) y[e] = (half_t)rounded_f32(__builtin_fmaf(x[e], s, (float)r[e]));
  *reinterpret_cast<v8h*>(out + o) = y;
}

// tanh for the GELUs: the result is rounded to fp16 (relative step 2^-11), so 1 - 2 / (1 + e^(2u)) on v_exp / v_rcp
// (absolute error ~1e-7) serves for |u| >= 0.1 and the odd series u - u^3/3 + 2u^5/15 (relative error < 6e-8) below;
// the device library's tanhf costs 4x the VALU work and made these kernels compute-bound (3.3 of 5.5 TB/s).
__device__ __forceinline__ float tanh_for_half(float u) {
  const float a = __builtin_fabsf(u);
  const float e = __builtin_amdgcn_exp2f(a * 2.8853900817779268f);          // e^(2|u|); inf for large |u| -> t = 1
  const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
  const float u2 = u * u;
  const float small = a * (1.0f + u2 * (-0.33333334f + u2 * 0.13333334f));
  const float t = a < 0.1f ? small : big;
  return __builtin_copysignf(t, u);
}

// gelu_new / gelu_fast in c10::Half arithmetic (every operator: f32 op, one rounding to fp16 -- which for + and * of
// two fp16 values equals the native fp16 instruction)
template <bool FAST>
__global__ __launch_bounds__(PT) void gelu_kernel(half_t* __restrict__ out, const half_t* __restrict__ in, int d) {
  const int i = (blockIdx.y * PT + threadIdx.x) * VT;
  if (i >= d) return;
  const size_t o = (size_t)blockIdx.x * d + i;
  const v8h x = *reinterpret_cast<const v8h*>(in + o);
  v8h u;      // the tanh argument, already rounded to fp16; packed fp16 instructions wherever both operands are fp16
  if constexpr (FAST) {
    v8h a, b;
#pragma unroll
    for (int e = 0; e < VT; ++e) { a[e] = (half_t)((float)x[e] * 0.79788456f); b[e] = (half_t)(0.044715f * (float)x[e]); }
    u = a * ((half_t)1.0f + b * x);
  } else {
    const v8h x3 = (x * x) * x;
    v8h k;
#pragma unroll
    for (int e = 0; e < VT; ++e) k[e] = (half_t)(0.044715f * (float)x3[e]);
    const v8h inner = x + k;
#pragma unroll
    for (int e = 0; e < VT; ++e) u[e] = (half_t)(0.79788456f * (float)inner[e]);
  }
  v8h th;
#pragma unroll
  for (int e = 0; e < VT; ++e) th[e] = (half_t)tanh_for_half((float)u[e]);
  *reinterpret_cast<v8h*>(out + o) = ((half_t)0.5f * x) * ((half_t)1.0f + th);
}

// silu(x) * y in f32 (activation_kernels.cu:10-13); v_exp / v_rcp stand in for the reference's fast-math expf / division
__device__ __forceinline__ float silu_mul_f32(float x, float y) {
  const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
  return rounded_f32(x * __builtin_amdgcn_rcpf(1.0f + e)) * y;
}

// invoke_dequant_silu_and_mul_quant, float scale_out: q = rni_sat(silu(gate * sg) * (up * su) / scale_out)
__global__ __launch_bounds__(PT) void dequant_silu_quant_static_kernel(int8_t* __restrict__ out,
                                                                        const int32_t* __restrict__ in, float sg,
                                                                        float su, float so, int d) {
  const int i = (blockIdx.y * PT + threadIdx.x) * VT;
  if (i >= d) return;
  const int32_t* row = in + (size_t)blockIdx.x * 2 * d;
  float g[VT], u[VT];
  load8_i32(row + i, g);
  load8_i32(row + d + i, u);
#pragma unroll
  for (int e = 0; e < VT; ++e) g[e] = rounded_f32(silu_mul_f32(rounded_f32(g[e] * sg), rounded_f32(u[e] * su))) / so;
  store8_codes(out + (size_t)blockIdx.x * d + i, g);
}

// invoke_dequant_silu_and_mul_quant, per token: tmp = t (f32), scale = amax / 127 (f32), q = rni_sat((127/amax) * t).
// One workgroup per token; the row's values stay in registers up to d = 256 * 8 * 8, beyond that they come back from tmp.
constexpr int SILU_RV = 8;
__global__ __launch_bounds__(PT) void dequant_silu_quant_token_kernel(int8_t* __restrict__ out,
                                                                       const int32_t* __restrict__ in, float sg,
                                                                       float su, float* __restrict__ scale_out,
                                                                       float* __restrict__ tmp, int d) {
  __shared__ float red[96];
  const int32_t* row = in + (size_t)blockIdx.x * 2 * d;
  float* trow = tmp + (size_t)blockIdx.x * d;
  float t[SILU_RV][VT];
  float amax = 0.0f;
#pragma unroll
  for (int it = 0; it < SILU_RV; ++it) {
    const int i = (threadIdx.x + it * PT) * VT;
    if (i < d) {
      float g[VT], u[VT];
      load8_i32(row + i, g);
      load8_i32(row + d + i, u);
#pragma unroll
      for (int e = 0; e < VT; ++e) {
        t[it][e] = rounded_f32(silu_mul_f32(rounded_f32(g[e] * sg), rounded_f32(u[e] * su)));
        amax = __builtin_fmaxf(amax, __builtin_fabsf(t[it][e]));
      }
      *reinterpret_cast<v4f*>(trow + i) = (v4f){t[it][0], t[it][1], t[it][2], t[it][3]};
      *reinterpret_cast<v4f*>(trow + i + 4) = (v4f){t[it][4], t[it][5], t[it][6], t[it][7]};
    }
  }
  for (int i = (threadIdx.x + SILU_RV * PT) * VT; i < d; i += PT * VT) {   // very long rows: values re-read from tmp
    float g[VT], u[VT];
    load8_i32(row + i, g);
    load8_i32(row + d + i, u);
    float v[VT];
#pragma unroll
    for (int e = 0; e < VT; ++e) {
      v[e] = rounded_f32(silu_mul_f32(rounded_f32(g[e] * sg), rounded_f32(u[e] * su)));
      amax = __builtin_fmaxf(amax, __builtin_fabsf(v[e]));
    }
    *reinterpret_cast<v4f*>(trow + i) = (v4f){v[0], v[1], v[2], v[3]};
    *reinterpret_cast<v4f*>(trow + i + 4) = (v4f){v[4], v[5], v[6], v[7]};
  }
  amax = block_max_rt<PT>(amax, red);
  if (threadIdx.x == 0) scale_out[blockIdx.x] = amax / 127.0f;
  const float q = 127.0f / amax;
  int8_t* orow = out + (size_t)blockIdx.x * d;
#pragma unroll
  for (int it = 0; it < SILU_RV; ++it) {
    const int i = (threadIdx.x + it * PT) * VT;
    if (i < d) {
      float c[VT];
#pragma unroll
      for (int e = 0; e < VT; ++e) c[e] = q * t[it][e];
      store8_codes(orow + i, c);
    }
  }
  for (int i = (threadIdx.x + SILU_RV * PT) * VT; i < d; i += PT * VT) {   // own stores: visible to this thread
    const v4f a = *reinterpret_cast<const v4f*>(trow + i), b = *reinterpret_cast<const v4f*>(trow + i + 4);
    float c[VT];
#pragma unroll
    for (int e = 0; e < 4; ++e) { c[e] = q * a[e]; c[4 + e] = q * b[e]; }
    store8_codes(orow + i, c);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// norms with a static output scale: one workgroup of 256 threads per token, the row parked as f32 in LDS (hidden <= 16256)
//   KIND 0  rms_norm(use_quant=True):  q = rni_sat((x * rstd) * w)
//   KIND 1  rms_norm_general, per-tensor scaling:  y = h(((x - mean) * rstd) * gamma), q = rni_sat(f32(y) * scale[0])
//   KIND 2  dequant_add_residual_rms_norm_quant:  diff = fma(acc, s, res) (f32, feeds the variance by FMA),
//           res <- h(diff), q = rni_sat((f32(h(diff)) * rstd) * gamma)
// ---------------------------------------------------------------------------------------------------------------
constexpr int NRV_MAX = 8;
template <int KIND, int NRV>      // NRV 16-B vectors per thread: hidden <= 256 * NRV * 8
__global__ __launch_bounds__(PT) void norm_static_kernel(int8_t* __restrict__ out, const void* __restrict__ in,
                                                          half_t* __restrict__ residual,
                                                          const half_t* __restrict__ gamma,
                                                          const half_t* __restrict__ scale_ptr, float scale, float eps,
                                                          int hidden, int nv) {
  extern __shared__ __attribute__((aligned(16))) float xs[];   // [hidden]
  __shared__ float red[96];
  const int p = threadIdx.x;
  const size_t row = (size_t)blockIdx.x * hidden;
  float x[NRV][VT];
  v8h g8[NRV];
  const float s = (KIND == 2 && scale_ptr) ? (float)scale_ptr[blockIdx.x] : scale;
#pragma unroll
  for (int it = 0; it < NRV; ++it) {
    const int i = (p + it * PT) * VT;
    if (i < hidden) {
      g8[it] = *reinterpret_cast<const v8h*>(gamma + i);
      if constexpr (KIND == 2) {
        float a[VT];
        load8_i32(reinterpret_cast<const int32_t*>(in) + row + i, a);
        const v8h r = *reinterpret_cast<const v8h*>(residual + row + i);
        v8h nr;
#pragma unroll
        for (int e = 0; e < VT; ++e) {
          x[it][e] = rounded_f32(__builtin_fmaf(a[e], s, (float)r[e]));
          nr[e] = (half_t)x[it][e];
        }
        *reinterpret_cast<v8h*>(residual + row + i) = nr;
      } else {
        const v8h t = *reinterpret_cast<const v8h*>(reinterpret_cast<const half_t*>(in) + row + i);
#pragma unroll
        for (int e = 0; e < VT; ++e) x[it][e] = (float)t[e];
      }
      *reinterpret_cast<v4f*>(xs + i) = (v4f){x[it][0], x[it][1], x[it][2], x[it][3]};
      *reinterpret_cast<v4f*>(xs + i + 4) = (v4f){x[it][4], x[it][5], x[it][6], x[it][7]};
    }
  }
  __syncthreads();
  float st[2][VT], tv[2];
  ordered_partials<2>(xs, p, nv, hidden, st, [](float (&v)[2][VT], int e, float val) {
    if (KIND == 1) v[0][e] = v[0][e] + val;
    v[1][e] = __builtin_fmaf(val, val, v[1][e]);   // fp16-valued inputs: val * val is exact, FMA == mul + add
  });
  tree_sum8<2>(st, red, p, nv >> 5, tv);
  const float mean = KIND == 1 ? tv[0] / (float)hidden : 0.0f;
  const float rstd = 1.0f / __builtin_sqrtf(tv[1] / (float)hidden + eps);
  const float so = KIND == 1 ? (float)scale_ptr[0] : 1.0f;
#pragma unroll
  for (int it = 0; it < NRV; ++it) {
    const int i = (p + it * PT) * VT;
    if (i < hidden) {
      float c[VT];
#pragma unroll
      for (int e = 0; e < VT; ++e) {
        if constexpr (KIND == 1) {
          const float y = rounded_f32((x[it][e] - mean) * rstd);
          c[e] = (float)(half_t)rounded_f32(y * (float)g8[it][e]) * so;
        } else {
          const float v = KIND == 2 ? (float)(half_t)x[it][e] : x[it][e];
          c[e] = rounded_f32(v * rstd) * (float)g8[it][e];
        }
      }
      store8_codes(out + row + i, c);
    }
  }
}

static inline bool row_shape_ok(int hidden, bool round32) {
  if (hidden < 8 || hidden % 8 || hidden > PT * NRV_MAX * VT || (size_t)hidden * sizeof(float) + 512 > 64 * 1024) return false;
  const int nv = hidden < 1024 ? hidden : 1024;
  return round32 || nv % 32 == 0;       // the un-rounded launches need whole virtual warps
}
#define OMNI_NORM_STATIC_LAUNCH(KIND, tokens, hidden, ...)                                                         \
  do {                                                                                                            \
    const size_t lds_ = (size_t)(hidden) * sizeof(float);                                                         \
    if ((hidden) <= PT * 2 * VT)                                                                                  \
      hipLaunchKernelGGL((norm_static_kernel<KIND, 2>), dim3(tokens), dim3(PT), lds_, (hipStream_t)stream, __VA_ARGS__); \
    else if ((hidden) <= PT * 4 * VT)                                                                             \
      hipLaunchKernelGGL((norm_static_kernel<KIND, 4>), dim3(tokens), dim3(PT), lds_, (hipStream_t)stream, __VA_ARGS__); \
    else                                                                                                          \
      hipLaunchKernelGGL((norm_static_kernel<KIND, 8>), dim3(tokens), dim3(PT), lds_, (hipStream_t)stream, __VA_ARGS__); \
  } while (0)
static inline int virtual_threads(int hidden) {
  const int nv = hidden < 1024 ? hidden : 1024;
  return 32 * ((nv + 31) / 32);
}

}  // namespace omni

using namespace omni;

#define OMNI_PT_GRID(tokens, cols) dim3((unsigned)(tokens), (unsigned)(((cols) / VT + PT - 1) / PT))

extern "C" int omni_quant_static(void* out_i8, const void* in_f16, float scale, int tokens, int hidden, void* stream) {
  if (!out_i8 || !in_f16 || tokens < 0 || hidden < 8 || hidden % 8) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  hipLaunchKernelGGL(quant_static_kernel, OMNI_PT_GRID(tokens, hidden), dim3(PT), 0, (hipStream_t)stream,
                     (int8_t*)out_i8, (const half_t*)in_f16, (float)(half_t)scale, hidden);
  return omni_launch_status();
}

extern "C" int omni_dequant(void* out_f16, const void* in_i32, float scale, int tokens, int hidden,
                            long long in_stride, long long out_stride, void* stream) {
  if (!out_f16 || !in_i32 || tokens < 0 || hidden < 8 || hidden % 8 || in_stride < hidden || out_stride < hidden ||
      in_stride % 4 || out_stride % 8)
    return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  hipLaunchKernelGGL(dequant_kernel, OMNI_PT_GRID(tokens, hidden), dim3(PT), 0, (hipStream_t)stream, (half_t*)out_f16,
                     (const int32_t*)in_i32, (float)(half_t)scale, hidden, in_stride, out_stride);
  return omni_launch_status();
}

extern "C" int omni_dequant_add_residual(void* out_f16, const void* in_i32, const void* residual_f16,
                                         const void* token_scale_f16, float scale, int tokens, int hidden,
                                         void* stream) {
  if (!out_f16 || !in_i32 || !residual_f16 || tokens < 0 || hidden < 8 || hidden % 8) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  hipLaunchKernelGGL(dequant_add_residual_kernel, OMNI_PT_GRID(tokens, hidden), dim3(PT), 0, (hipStream_t)stream,
                     (half_t*)out_f16, (const int32_t*)in_i32, (const half_t*)residual_f16,
                     (const half_t*)token_scale_f16, (float)(half_t)scale, hidden);
  return omni_launch_status();
}

extern "C" int omni_gelu(void* out_f16, const void* in_f16, int kind, int tokens, int d, void* stream) {
  if (!out_f16 || !in_f16 || tokens < 0 || d < 8 || d % 8 || (kind != 0 && kind != 1)) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  if (kind == 0)
    hipLaunchKernelGGL(gelu_kernel<false>, OMNI_PT_GRID(tokens, d), dim3(PT), 0, (hipStream_t)stream,
                       (half_t*)out_f16, (const half_t*)in_f16, d);
  else
    hipLaunchKernelGGL(gelu_kernel<true>, OMNI_PT_GRID(tokens, d), dim3(PT), 0, (hipStream_t)stream,
                       (half_t*)out_f16, (const half_t*)in_f16, d);
  return omni_launch_status();
}

extern "C" int omni_dequant_silu_and_mul_quant(void* out_i8, const void* in_i32, float scale_gate, float scale_up,
                                               float scale_out, void* token_scale_f32, void* tmp_f32, int tokens,
                                               int d, void* stream) {
  if (!out_i8 || !in_i32 || tokens < 0 || d < 8 || d % 8) return OMNI_EINVAL;
  if ((token_scale_f32 == nullptr) != (tmp_f32 == nullptr)) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  if (token_scale_f32)
    hipLaunchKernelGGL(dequant_silu_quant_token_kernel, dim3(tokens), dim3(PT), 0, (hipStream_t)stream,
                       (int8_t*)out_i8, (const int32_t*)in_i32, scale_gate, scale_up, (float*)token_scale_f32,
                       (float*)tmp_f32, d);
  else
    hipLaunchKernelGGL(dequant_silu_quant_static_kernel, OMNI_PT_GRID(tokens, d), dim3(PT), 0, (hipStream_t)stream,
                       (int8_t*)out_i8, (const int32_t*)in_i32, scale_gate, scale_up, scale_out, d);
  return omni_launch_status();
}

extern "C" int omni_rms_norm_quant(void* out_i8, const void* in_f16, const void* weight_f16, float eps, int tokens,
                                   int hidden, void* stream) {
  if (!out_i8 || !in_f16 || !weight_f16 || tokens < 0 || !row_shape_ok(hidden, false)) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  OMNI_NORM_STATIC_LAUNCH(0, tokens, hidden, (int8_t*)out_i8, in_f16, (half_t*)nullptr, (const half_t*)weight_f16,
                          (const half_t*)nullptr, 1.0f, eps, hidden, virtual_threads(hidden));
  return omni_launch_status();
}

extern "C" int omni_rms_norm_general_static(void* out_i8, const void* in_f16, const void* weight_f16,
                                            const void* scaling_f16, float eps, int tokens, int hidden, void* stream) {
  if (!out_i8 || !in_f16 || !weight_f16 || !scaling_f16 || tokens < 0 || !row_shape_ok(hidden, true))
    return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  OMNI_NORM_STATIC_LAUNCH(1, tokens, hidden, (int8_t*)out_i8, in_f16, (half_t*)nullptr, (const half_t*)weight_f16,
                          (const half_t*)scaling_f16, 1.0f, eps, hidden, virtual_threads(hidden));
  return omni_launch_status();
}

extern "C" int omni_dequant_add_residual_rms_norm_quant(void* out_i8, const void* in_i32, void* residual_f16,
                                                        const void* gamma_f16, const void* token_scale_f16,
                                                        float scale, float eps, int tokens, int hidden, void* stream) {
  if (!out_i8 || !in_i32 || !residual_f16 || !gamma_f16 || tokens < 0 || !row_shape_ok(hidden, false))
    return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  OMNI_NORM_STATIC_LAUNCH(2, tokens, hidden, (int8_t*)out_i8, in_i32, (half_t*)residual_f16, (const half_t*)gamma_f16,
                          (const half_t*)token_scale_f16, (float)(half_t)scale, eps, hidden, virtual_threads(hidden));
  return omni_launch_status();
}
