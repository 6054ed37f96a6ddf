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

// Fused extension: gate_up projection + silu_and_mul in one kernel (act fp16 [M, N/2]) + row maxima of |act|.
extern "C" int omni_w4a8_per_chn_gemm_silu(const void* in_feats, const void* qweight, const void* wscales,
                                           const void* ascales, const void* w_szs, const void* a_ssums, void* act_f16,
                                           void* amax_slots_u32, int M, int N, int K, void* stream) {
  if (!in_feats || !qweight || !wscales || !ascales || !w_szs || !a_ssums || !act_f16 || !amax_slots_u32) return OMNI_EINVAL;
  GemmArgs a{};
  a.A = (const int8_t*)in_feats; a.W = (const uint8_t*)qweight;
  a.wscales = (const half_t*)wscales; a.ascales = (const half_t*)ascales;
  a.wsz = (const half_t*)w_szs; a.asum = (const half_t*)a_ssums;
  a.out = (half_t*)act_f16; a.M = M; a.N = N; a.K = K; a.out_stride = N / 2;
  a.amax = (uint32_t*)amax_slots_u32;
  return launch_gemm_silu<MODE_CHN>(a, (hipStream_t)stream);
}

// Fused extension: split-K partial sums of a projection whose int8 input is quantised on the fly from fp16 activations.
extern "C" int omni_w4a8_per_chn_gemm_partial_f16(const void* act_f16, const void* amax_slots_u32, const void* qweight,
                                                  void* slab_i32, size_t slab_bytes, void* sum_f16, void* scale_f16,
                                                  int M, int N, int K, int* sk_out, void* stream) {
  if (!act_f16 || !amax_slots_u32 || !qweight || !scale_f16) return OMNI_EINVAL;
  GemmArgs a{};
  a.A16 = (const half_t*)act_f16; a.amax = (uint32_t*)amax_slots_u32; a.W = (const uint8_t*)qweight;
  a.sum_out = (half_t*)sum_f16; a.scale_out = (half_t*)scale_f16;
  a.M = M; a.N = N; a.K = K; a.out_stride = N;
  return launch_gemm_partial_f16<MODE_CHN>(a, slab_i32, slab_bytes, sk_out, (hipStream_t)stream);
}

OMNI_CLK_READER(omni_debug_clocks_gemm_chn)
#ifdef OMNI_DEBUG_CLOCKS
// timeline probe of the mid-M kernel (tools/midm_timeline.py): 2 workgroups x 8 waves x 104 stamps
extern "C" int omni_debug_timeline_midm_chn(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(omni::omni_dbg_midm), sizeof(omni::omni_dbg_midm)) == hipSuccess ? 0 : -5;
}
// timeline probe of the exact prefill kernel (tools/gemm_timeline.py): 4 marks per workgroup
extern "C" int omni_debug_timeline_gemm_chn(unsigned long long* out, int nwg) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(omni::omni_dbg_tl), (size_t)nwg * 5 * sizeof(unsigned long long)) == hipSuccess ? 0 : -5;
}
#endif
