// omni_w8a8_gemm: instantiates the W8A8 kernels (see qgemm_kernel.h).
#include "qgemm_kernel.h"
using namespace omni;

extern "C" int omni_w8a8_gemm(const void* in_feats, const void* weight, const void* wscales,
                              const void* ascales, void* out_feats, int M, int N, int K,
                              int64_t out_row_stride, void* workspace, size_t workspace_bytes,
                              void* stream) {
  if (!in_feats || !weight || !wscales || !ascales || !out_feats) return OMNI_EINVAL;
  GemmArgs a{};
  a.A = (const int8_t*)in_feats; a.W = (const uint8_t*)weight;
  a.wscales = (const half_t*)wscales; a.ascales = (const half_t*)ascales;
  a.out = (half_t*)out_feats; a.M = M; a.N = N; a.K = K; a.out_stride = out_row_stride;
  return launch_gemm<MODE_W8>(a, workspace, workspace_bytes, (hipStream_t)stream);
}

// Fused extension: split-K partial sums only (see omni_splitk_w8_add_rms_norm_general_fuse_sum).
extern "C" int omni_w8a8_gemm_partial(const void* in_feats, const void* weight, void* slab_i32, size_t slab_bytes, int M,
                                      int N, int K, int* sk_out, void* stream) {
  if (!in_feats || !weight) return OMNI_EINVAL;
  GemmArgs a{};
  a.A = (const int8_t*)in_feats; a.W = (const uint8_t*)weight;
  a.M = M; a.N = N; a.K = K; a.out_stride = N;
  return launch_gemm_partial<MODE_W8>(a, slab_i32, slab_bytes, sk_out, (hipStream_t)stream);
}

// Fused extension: gate_up projection + silu_and_mul in one kernel (act fp16 [M, N/2]) + row maxima of |act|: the W8A8
// form of omni_w4a8_per_chn_gemm_silu (gate_up_proj -> act_fn, llama_w8a8_unpad.py:94-105).  M <= 16.
extern "C" int omni_w8a8_gemm_silu(const void* in_feats, const void* weight, const void* wscales, const void* ascales,
                                   void* act_f16, void* amax_slots_u32, int M, int N, int K, void* stream) {
  if (!in_feats || !weight || !wscales || !ascales || !act_f16 || !amax_slots_u32) return OMNI_EINVAL;
  GemmArgs a{};
  a.A = (const int8_t*)in_feats; a.W = (const uint8_t*)weight;
  a.wscales = (const half_t*)wscales; a.ascales = (const half_t*)ascales;
  a.out = (half_t*)act_f16; a.M = M; a.N = N; a.K = K; a.out_stride = N / 2;
  a.amax = (uint32_t*)amax_slots_u32;
  return launch_gemm_silu<MODE_W8>(a, (hipStream_t)stream);
}

// Fused extension: split-K partial sums of a W8A8 projection whose int8 input is quantised on the fly from fp16
// activations (invoke_quant's arithmetic) with the row maxima its producer left; rider workgroups write the scales.
extern "C" int omni_w8a8_gemm_partial_f16(const void* act_f16, const void* amax_slots_u32, const void* weight,
                                          void* slab_i32, size_t slab_bytes, void* scale_f16, int M, int N, int K,
                                          int* sk_out, void* stream) {
  if (!act_f16 || !amax_slots_u32 || !weight || !scale_f16) return OMNI_EINVAL;
  GemmArgs a{};
  a.A16 = (const half_t*)act_f16; a.amax = (uint32_t*)amax_slots_u32; a.W = (const uint8_t*)weight;
  a.sum_out = nullptr; a.scale_out = (half_t*)scale_f16;
  a.M = M; a.N = N; a.K = K; a.out_stride = N;
  return launch_gemm_partial_f16<MODE_W8>(a, slab_i32, slab_bytes, sk_out, (hipStream_t)stream);
}
