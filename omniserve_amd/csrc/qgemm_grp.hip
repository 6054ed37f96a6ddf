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
