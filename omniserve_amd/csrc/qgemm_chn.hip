// omni_w4a8_per_chn_gemm: instantiates the per-channel W4A8 kernels (see qgemm_kernel.h).
#include "qgemm_kernel.h"
using namespace omni;

extern "C" int omni_w4a8_per_chn_gemm(const void* in_feats, const void* qweight, const void* wscales,
                                      const void* ascales, const void* w_szs, const void* a_ssums,
                                      void* out_feats, int M, int N, int K, int64_t out_row_stride,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  if (!in_feats || !qweight || !wscales || !ascales || !w_szs || !a_ssums || !out_feats) return OMNI_EINVAL;
  GemmArgs a{};
  a.A = (const int8_t*)in_feats; a.W = (const uint8_t*)qweight;
  a.wscales = (const half_t*)wscales; a.ascales = (const half_t*)ascales;
  a.wsz = (const half_t*)w_szs; a.asum = (const half_t*)a_ssums;
  a.out = (half_t*)out_feats; a.M = M; a.N = N; a.K = K; a.out_stride = out_row_stride;
  return launch_gemm<MODE_CHN>(a, workspace, workspace_bytes, (hipStream_t)stream);
}

// Fused extension: split-K partial sums only (see omni_splitk_add_rms_norm_general_fuse_sum).
extern "C" int omni_w4a8_per_chn_gemm_partial(const void* in_feats, const void* qweight, void* slab_i32,
                                              size_t slab_bytes, int M, int N, int K, int* sk_out, void* stream) {
  if (!in_feats || !qweight) return OMNI_EINVAL;
  GemmArgs a{};
  a.A = (const int8_t*)in_feats; a.W = (const uint8_t*)qweight;
  a.M = M; a.N = N; a.K = K; a.out_stride = N;
  return launch_gemm_partial<MODE_CHN>(a, slab_i32, slab_bytes, sk_out, (hipStream_t)stream);
}

OMNI_CLK_READER(omni_debug_clocks_gemm_chn)
