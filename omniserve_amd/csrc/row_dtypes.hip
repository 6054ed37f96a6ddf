// bf16 / fp32 instantiations of the row kernels the reference dispatches over float, half and bfloat16
// (VLLM_DISPATCH_FLOATING_TYPES, kernels/csrc/dispatch_utils.h:7-14): invoke_quant[_fuse_sum] (fused_kernels.cu:212-272),
// rms_norm (layernorm_kernels.cu:408-431), rms_norm_general[_fuse_sum] (:433-513), silu_and_mul (activation_kernels.cu:84-97).
// The Llama W4A8 / W8A8 paths run fp16 and use the tuned kernels of elementwise.hip; these cover the other two element
// types with the reference's own geometry -- one workgroup per token, block = min(hidden, 1024) (rounded up to 32 for the
// general norm), thread t accumulating elements t, t + block, ... -- and the reference's rounding points, which depend on T:
// the general norm rounds its output, its running maximum and its per-thread running sum to T (`T_scalar amax`, `T_scalar
// sum`, layernorm_kernels.cu:279-292), rms_norm and silu_and_mul round their intermediate and their result to T (c10 scalar
// types multiply in float and round back).  Scales and sums stay fp16 (`at::Half`) for every T.  Bit-exact vs
// oracle/elementwise.py (dtype = "bf16" / "f32"); SiLU: bf16 uses v_exp / v_rcp like the fp16 kernel (<= 2 ulp of T), fp32 the
// library exp and an IEEE division (<= 4 ulp of the oracle's float32 evaluation).
#include "common.h"

namespace omni {

enum { DT_F16 = 0, DT_BF16 = 1, DT_F32 = 2 };

struct bf16_t { uint16_t bits; };
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __builtin_bit_cast(float, (uint32_t)v.bits << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {       // round to nearest even; NaN stays NaN
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return bf16_t{(uint16_t)((u >> 16) | 0x0040u)};
  return bf16_t{(uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16)};
}
template <typename T> struct El;
template <> struct El<bf16_t> {
  static __device__ __forceinline__ float load(const bf16_t* p, size_t i) { return bf16_to_f32(p[i]); }
  static __device__ __forceinline__ float round(float f) { return bf16_to_f32(f32_to_bf16(f)); }
  static __device__ __forceinline__ void store(bf16_t* p, size_t i, float f) { p[i] = f32_to_bf16(f); }
};
template <> struct El<float> {
  static __device__ __forceinline__ float load(const float* p, size_t i) { return p[i]; }
  static __device__ __forceinline__ float round(float f) { return f; }
  static __device__ __forceinline__ void store(float* p, size_t i, float f) { p[i] = f; }
};

// invoke_quant / invoke_quant_fuse_sum: T only enters through the load (fused_kernels.cu:57-142)
template <typename T, bool FUSE_SUM>
__global__ __launch_bounds__(1024) void quant_dt_kernel(int8_t* __restrict__ out, const T* __restrict__ in,
                                                         half_t* __restrict__ sum_out, half_t* __restrict__ scale_out, int hidden) {
  __shared__ float red[32];
  const int tid = threadIdx.x, nt = blockDim.x;
  const size_t row = (size_t)blockIdx.x * hidden;
  float amax = 0.0f, s = 0.0f;
  for (int i = tid; i < hidden; i += nt) {
    const float v = El<T>::load(in, row + i);
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
  for (int i = tid; i < hidden; i += nt) out[row + i] = rni_sat_s8(El<T>::load(in, row + i) * q);
}

// generalLayerNorm[_fuse_sum]<T, at::Half>, per-token path (layernorm_kernels.cu:58-331)
template <typename T, bool FUSE_SUM>
__global__ __launch_bounds__(1024) void general_norm_dt_kernel(int8_t* __restrict__ out, const T* __restrict__ in,
                                                                const T* __restrict__ gamma, half_t* __restrict__ sum_out,
                                                                half_t* __restrict__ scale_out, float eps, int hidden) {
  __shared__ float red[32];
  const int tid = threadIdx.x, nt = blockDim.x;
  const size_t row = (size_t)blockIdx.x * hidden;
  float lsum = 0.0f, lsq = 0.0f;
  for (int i = tid; i < hidden; i += nt) {
    const float v = El<T>::load(in, row + i);
    lsum = lsum + v;
    lsq = lsq + v * v;
  }
  const float mean = ref_block_sum(lsum, red) / (float)hidden;
  const float var = ref_block_sum(lsq, red);
  const float rstd = 1.0f / __builtin_sqrtf(var / (float)hidden + eps);
  float amax_t = El<T>::round(1e-6f);
  float tsum = 0.0f;
  for (int i = tid; i < hidden; i += nt) {
    float y = (El<T>::load(in, row + i) - mean) * rstd;
    y = rounded_f32(y * El<T>::load(gamma, i));
    const float yt = El<T>::round(y);
    amax_t = __builtin_fmaxf(amax_t, __builtin_fabsf(yt));
    if constexpr (FUSE_SUM) tsum = El<T>::round(tsum + yt);
  }
  const float amax = ref_block_max(amax_t, red, -1e20f);
  if constexpr (FUSE_SUM) {
    const float tot = ref_block_sum(tsum, red);
    if (tid == 0) sum_out[blockIdx.x] = (half_t)tot;
  }
  if (tid == 0) scale_out[blockIdx.x] = (half_t)(amax / 127.0f);
  const float q = 127.0f / amax;
  for (int i = tid; i < hidden; i += nt) {
    float y = (El<T>::load(in, row + i) - mean) * rstd;
    y = rounded_f32(y * El<T>::load(gamma, i));
    out[row + i] = rni_sat_s8(y * q);
  }
}

// rms_norm_kernel<T, T, false> (layernorm_kernels.cu:335-365): out = T( f32(T(x * rstd)) * f32(w) )
template <typename T>
__global__ __launch_bounds__(1024) void rms_norm_dt_kernel(T* __restrict__ out, const T* __restrict__ in, const T* __restrict__ weight,
                                                            float eps, int hidden) {
  __shared__ float red[32];
  const int tid = threadIdx.x, nt = blockDim.x;
  const size_t row = (size_t)blockIdx.x * hidden;
  float lsq = 0.0f;
  for (int i = tid; i < hidden; i += nt) {
    const float v = El<T>::load(in, row + i);
    lsq = lsq + v * v;
  }
  const float var = ref_block_sum(lsq, red);
  const float rstd = 1.0f / __builtin_sqrtf(var / (float)hidden + eps);
  for (int i = tid; i < hidden; i += nt) {
    const float t = El<T>::round(rounded_f32(El<T>::load(in, row + i) * rstd));
    El<T>::store(out, row + i, rounded_f32(t * El<T>::load(weight, i)));
  }
}

// silu_and_mul_kernel<T> (activation_kernels.cu:10-30): out = T( f32(T(x / (1 + exp(-x)))) * f32(y) )
template <typename T>
__global__ __launch_bounds__(256) void silu_and_mul_dt_kernel(T* __restrict__ out, const T* __restrict__ in, int tokens, int d) {
  const size_t total = (size_t)tokens * d;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t t = idx / d;
    const int c = (int)(idx % d);
    const float x = El<T>::load(in, t * 2 * d + c), y = El<T>::load(in, t * 2 * d + d + c);
    float sf;
    if constexpr (sizeof(T) == 4) {      // fp32: the approximate pair below is 4-5 ulp of f32 off; library exp + IEEE division
      sf = x / (1.0f + expf(-x));
    } else {
      const float e = __builtin_amdgcn_exp2f(-x * 1.4426950408889634f);
      sf = x * __builtin_amdgcn_rcpf(1.0f + e);
    }
    const float s = El<T>::round(rounded_f32(sf));
    El<T>::store(out, idx, rounded_f32(s * y));
  }
}

static inline int block_for(int hidden, bool round32) {
  int b = hidden < 1024 ? hidden : 1024;
  if (round32) b = 32 * ((b + 31) / 32);
  return b;
}

}  // namespace omni
using namespace omni;

#define OMNI_DT_SWITCH(dtype, CALL)                       \
  switch (dtype) {                                        \
    case DT_BF16: { typedef bf16_t T; CALL; break; }      \
    case DT_F32: { typedef float T; CALL; break; }        \
    default: return OMNI_EINVAL;                          \
  }

extern "C" int omni_quant_dt(void* out_i8, const void* in, void* sum_f16_or_null, void* scale_f16, int tokens, int hidden,
                             int dtype, void* stream) {
  if (!out_i8 || !in || !scale_f16 || tokens < 0 || hidden < 1 || hidden % 32 != 0) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  const dim3 g(tokens), b(block_for(hidden, false));
  if (sum_f16_or_null) {
    OMNI_DT_SWITCH(dtype, hipLaunchKernelGGL((quant_dt_kernel<T, true>), g, b, 0, (hipStream_t)stream, (int8_t*)out_i8, (const T*)in,
                                             (half_t*)sum_f16_or_null, (half_t*)scale_f16, hidden))
  } else {
    OMNI_DT_SWITCH(dtype, hipLaunchKernelGGL((quant_dt_kernel<T, false>), g, b, 0, (hipStream_t)stream, (int8_t*)out_i8, (const T*)in,
                                             (half_t*)nullptr, (half_t*)scale_f16, hidden))
  }
  return omni_launch_status();
}

extern "C" int omni_rms_norm_general_dt(void* out_i8, const void* in, const void* weight, void* sum_f16_or_null, void* scale_f16,
                                        float eps, int tokens, int hidden, int dtype, void* stream) {
  if (!out_i8 || !in || !weight || !scale_f16 || tokens < 0 || hidden < 1) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  const dim3 g(tokens), b(block_for(hidden, true));
  if (sum_f16_or_null) {
    OMNI_DT_SWITCH(dtype, hipLaunchKernelGGL((general_norm_dt_kernel<T, true>), g, b, 0, (hipStream_t)stream, (int8_t*)out_i8,
                                             (const T*)in, (const T*)weight, (half_t*)sum_f16_or_null, (half_t*)scale_f16, eps, hidden))
  } else {
    OMNI_DT_SWITCH(dtype, hipLaunchKernelGGL((general_norm_dt_kernel<T, false>), g, b, 0, (hipStream_t)stream, (int8_t*)out_i8,
                                             (const T*)in, (const T*)weight, (half_t*)nullptr, (half_t*)scale_f16, eps, hidden))
  }
  return omni_launch_status();
}

extern "C" int omni_rms_norm_dt(void* out, const void* in, const void* weight, float eps, int tokens, int hidden, int dtype,
                                void* stream) {
  if (!out || !in || !weight || tokens < 0 || hidden < 1 || hidden % 32 != 0) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  OMNI_DT_SWITCH(dtype, hipLaunchKernelGGL((rms_norm_dt_kernel<T>), dim3(tokens), dim3(block_for(hidden, false)), 0,
                                           (hipStream_t)stream, (T*)out, (const T*)in, (const T*)weight, eps, hidden))
  return omni_launch_status();
}

extern "C" int omni_silu_and_mul_dt(void* out, const void* in, int tokens, int d, int dtype, void* stream) {
  if (!out || !in || tokens < 0 || d < 1) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  const size_t total = (size_t)tokens * d;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  OMNI_DT_SWITCH(dtype, hipLaunchKernelGGL((silu_and_mul_dt_kernel<T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (T*)out,
                                           (const T*)in, tokens, d))
  return omni_launch_status();
}
