// omni_w4a8_per_group_gemm: instantiates the g128 per-group W4A8 kernels (see qgemm_kernel.h).
#include "qgemm_kernel.h"
using namespace omni;

extern "C" int omni_w4a8_per_group_gemm(const void* in_feats, const void* qweight, const void* zeros,
                                        const void* scales_i8, const void* wscales, const void* ascales,
                                        void* out_feats, int M, int N, int K, int64_t out_row_stride,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  if (!in_feats || !qweight || !zeros || !scales_i8 || !wscales || !ascales || !out_feats) return OMNI_EINVAL;
  GemmArgs a{};
  a.A = (const int8_t*)in_feats; a.W = (const uint8_t*)qweight;
  a.s2s = (const uint8_t*)scales_i8; a.s2z = (const uint8_t*)zeros;
  a.wscales = (const half_t*)wscales; a.ascales = (const half_t*)ascales;
  a.out = (half_t*)out_feats; a.M = M; a.N = N; a.K = K; a.out_stride = out_row_stride;
  return launch_gemm<MODE_GRP>(a, workspace, workspace_bytes, (hipStream_t)stream);
}

// Fused extension: split-K partial sums only; the consumer (omni_splitk_w8_add_rms_norm_general_fuse_sum: the per-group
// epilogue h(f32(acc) * (wscales[n] * ascales[m])) is the W8A8 one) reduces the slabs.
extern "C" int omni_w4a8_per_group_gemm_partial(const void* in_feats, const void* qweight, const void* zeros,
                                                const void* scales_i8, void* slab_i32, size_t slab_bytes, int M, int N,
                                                int K, int* sk_out, void* stream) {
  if (!in_feats || !qweight || !zeros || !scales_i8) return OMNI_EINVAL;
  GemmArgs a{};
  a.A = (const int8_t*)in_feats; a.W = (const uint8_t*)qweight;
  a.s2s = (const uint8_t*)scales_i8; a.s2z = (const uint8_t*)zeros;
  a.M = M; a.N = N; a.K = K; a.out_stride = N;
  return launch_gemm_partial<MODE_GRP>(a, slab_i32, slab_bytes, sk_out, (hipStream_t)stream);
}

// Fused extension (g128): gate_up + silu_and_mul in one kernel; projection from fp16 activations (see qgemm_chn.hip).
extern "C" int omni_w4a8_per_group_gemm_silu(const void* in_feats, const void* qweight, const void* zeros,
                                             const void* scales_i8, const void* wscales, const void* ascales,
                                             void* act_f16, void* amax_slots_u32, int M, int N, int K, void* stream) {
  if (!in_feats || !qweight || !zeros || !scales_i8 || !wscales || !ascales || !act_f16 || !amax_slots_u32) return OMNI_EINVAL;
  GemmArgs a{};
  a.A = (const int8_t*)in_feats; a.W = (const uint8_t*)qweight;
  a.s2s = (const uint8_t*)scales_i8; a.s2z = (const uint8_t*)zeros;
  a.wscales = (const half_t*)wscales; a.ascales = (const half_t*)ascales;
  a.out = (half_t*)act_f16; a.M = M; a.N = N; a.K = K; a.out_stride = N / 2;
  a.amax = (uint32_t*)amax_slots_u32;
  return launch_gemm_silu<MODE_GRP>(a, (hipStream_t)stream);
}

extern "C" int omni_w4a8_per_group_gemm_partial_f16(const void* act_f16, const void* amax_slots_u32, const void* qweight,
                                                    const void* zeros, const void* scales_i8, void* slab_i32,
                                                    size_t slab_bytes, void* sum_f16, void* scale_f16, int M, int N, int K,
                                                    int* sk_out, void* stream) {
  if (!act_f16 || !amax_slots_u32 || !qweight || !zeros || !scales_i8 || !scale_f16) return OMNI_EINVAL;
  GemmArgs a{};
  a.A16 = (const half_t*)act_f16; a.amax = (uint32_t*)amax_slots_u32; a.W = (const uint8_t*)qweight;
  a.s2s = (const uint8_t*)scales_i8; a.s2z = (const uint8_t*)zeros;
  a.sum_out = (half_t*)sum_f16; a.scale_out = (half_t*)scale_f16;
  a.M = M; a.N = N; a.K = K; a.out_stride = N;
  return launch_gemm_partial_f16<MODE_GRP>(a, slab_i32, slab_bytes, sk_out, (hipStream_t)stream);
}
