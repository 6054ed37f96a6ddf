// omni_w4a8_per_{chn,group}_norm_gemm_fused: a (residual add + rms_norm_general[_fuse_sum]) -> (W4A8 GEMV) pair of the decode
// layer as ONE launch (norm_gemv_fused.h).  Replaces, bit for bit, omni_splitk_[w8_]add_rms_norm_general_fuse_sum (or
// omni_add_rms_norm_general_fuse_sum) followed by omni_w4a8_per_*_gemm / omni_w4a8_per_*_gemm_silu -- i.e. the reference's
// input_layernorm -> qkv_proj and post_attention_layernorm -> gate_up_proj (+ silu_and_mul) edges, llama_w4a8_unpad.py:410-432.
#include "norm_gemv_fused.h"
using namespace omni;

namespace {

struct SrcAddB {  // residual += delta (fp16 add), in place (elementwise.hip's SrcAdd)
  static constexpr bool BATCH = true;
  struct Raw { v8h a, d; };
  half_t* res;
  const half_t* delta;
  int stride;
  __device__ __forceinline__ void pin() const { asm volatile("" ::"s"(res), "s"(delta), "s"(stride)); }
  __device__ __forceinline__ SrcAddB at_row(int m) const { return SrcAddB{res + (size_t)m * stride, delta + (size_t)m * stride, stride}; }
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
struct SrcPlainB {  // the residual row as it is (first layer: the embedding rows)
  static constexpr bool BATCH = true;
  struct Raw { v8h t; };
  const half_t* row;
  int stride;
  __device__ __forceinline__ void pin() const { asm volatile("" ::"s"(row), "s"(stride)); }
  __device__ __forceinline__ SrcPlainB at_row(int m) const { return SrcPlainB{row + (size_t)m * stride, stride}; }
  __device__ __forceinline__ void fetch(int i, Raw& r) const { r.t = *reinterpret_cast<const v8h*>(row + i); }
  __device__ __forceinline__ void finish(int i, const Raw& r, float (&x)[VT]) const {
#pragma unroll
    for (int e = 0; e < VT; ++e) x[e] = (float)r.t[e];
  }
};

template <int MODE_, typename S> struct Tagged { static constexpr int MODE_OF = MODE_; typedef S Src; S s; };

int norm_block32(int hidden) {
  int b = hidden < 1024 ? hidden : 1024;
  return 32 * ((b + 31) / 32);
}

// every workgroup of the grid resident at once: the rows can never wait for a CU behind polling tiles
template <typename K>
bool grid_is_resident(K kern, int grid) {
  int dev = 0, per_cu = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, NGF_THREADS, 0) != hipSuccess) return false;
  // (MI355X_MICROARCH.md: the query can be one block per CU high at 82+ SGPRs; this kernel is bounded by its 64 KiB of LDS
  //  -- two per CU -- and by __launch_bounds__(256, 2); cap at that)
  if (per_cu > 2) per_cu = 2;
  return (long long)per_cu * cus >= grid;
}

template <int MODE, int EPI, int RV, bool FUSE_SUM, typename Src>
int launch_one(const NormGemvArgs& a, const Src& src, int grid, hipStream_t st) {
  auto kern = norm_gemv_fused_kernel<MODE, EPI, RV, FUSE_SUM, Src>;
  static thread_local int ok_grid = 0;      // largest grid checked resident (per kernel instantiation and thread)
  if (grid > ok_grid) {
    if (!grid_is_resident(kern, grid)) return OMNI_EINVAL;
    ok_grid = grid;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NGF_THREADS), 0, st, a, src);
  return omni_launch_status();
}

template <int MODE, bool FUSE_SUM, typename Src>
int launch_src(const NormGemvArgs& a, const Src& src, hipStream_t st) {
  const bool silu = a.amax != nullptr;
  const int grid = a.M + a.N / 64;      // (SiLU form: tile = tile row g of the gate half + tile row g of the up half)
  const bool rv2 = a.K > NGF_THREADS * VT;
  if (silu) return rv2 ? launch_one<MODE, 1, 2, FUSE_SUM, Src>(a, src, grid, st) : launch_one<MODE, 1, 1, FUSE_SUM, Src>(a, src, grid, st);
  return rv2 ? launch_one<MODE, 0, 2, FUSE_SUM, Src>(a, src, grid, st) : launch_one<MODE, 0, 1, FUSE_SUM, Src>(a, src, grid, st);
}

bool shape_ok(int M, int N, int K, int mode, bool silu) {
  if (M < 1 || M > 16 || N < 64 || K < 256) return false;
  const int kalign = mode == MODE_GRP ? 128 : 64;
  if (K % (NGF_KW * kalign) != 0 || K / NGF_KW > NGF_RING * KSTEP) return false;   // whole part in the register ring
  if (K > NGF_THREADS * 2 * VT) return false;                                       // row geometry <256, 2>
  if (N % (silu ? 128 : 64) != 0) return false;
  return true;
}

}  // namespace

// 1: the (M, N, K) pair form exists (mode 0 per-channel, 1 per-group; silu: the gate_up form) AND its grid is resident on
// this device; 0: use the two launches.
extern "C" int omni_norm_gemm_fused_ok(int M, int N, int K, int mode, int silu) {
  if (mode != MODE_CHN && mode != MODE_GRP) return 0;
  if (!shape_ok(M, N, K, mode, silu != 0)) return 0;
  const int grid = M + N / 64;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return 2LL * cus >= grid ? 1 : 0;
}

// Source of the rows (exactly one of the three):
//   slab_i32 != NULL  residual += h(epilogue(sum of sk split-K slabs))       = omni_splitk_[w8_]add_rms_norm_general_fuse_sum
//                     (p_wszs / p_asums NULL: the per-group / W8A8 epilogue acc * (sw * sa))
//   delta_f16 != NULL residual += delta                                      = omni_add_rms_norm_general_fuse_sum
//   neither           the residual as it is                                  = omni_rms_norm_general[_fuse_sum]
// GEMV: amax_slots_u32 != NULL selects the gate_up form (out = act [M, N/2], omni_w4a8_per_*_gemm_silu); sum_f16 NULL: no row
// sum (per-group).  sync_u32: NGF_SYNC_WORDS words ZEROED by the caller before the launch; err_u32: sticky error word
// (non-zero after a launch whose rows never arrived: the outputs are NaN).
static int norm_gemm_fused(int mode, void* codes_i8, void* residual_f16, const void* slab_i32, int sk, const void* delta_f16,
                           const void* p_wscales, const void* p_ascales, const void* p_wszs, const void* p_asums,
                           const void* gamma_f16, void* sum_f16, void* scale_f16, float eps, const void* qweight,
                           const void* zeros_i8, const void* scales_i8, const void* wscales, const void* w_szs, void* out_f16,
                           long long out_row_stride, void* amax_slots_u32, void* sync_u32, void* err_u32, int M, int N, int K,
                           void* clk, void* stream) {
  if (!codes_i8 || !residual_f16 || !gamma_f16 || !scale_f16 || !qweight || !wscales || !out_f16 || !sync_u32 || !err_u32)
    return OMNI_EINVAL;
  if (mode == MODE_CHN && (!w_szs || !sum_f16)) return OMNI_EINVAL;
  if (mode == MODE_GRP && (!zeros_i8 || !scales_i8)) return OMNI_EINVAL;
  if (slab_i32 && (sk < 1 || !p_wscales || !p_ascales || ((p_wszs == nullptr) != (p_asums == nullptr)))) return OMNI_EINVAL;
  if (!shape_ok(M, N, K, mode, amax_slots_u32 != nullptr)) return OMNI_EINVAL;
  NormGemvArgs a{};
  a.gamma = (const half_t*)gamma_f16; a.codes = (int8_t*)codes_i8; a.sum_out = (half_t*)sum_f16; a.scale_out = (half_t*)scale_f16;
  a.eps = eps; a.nv = norm_block32(K);
  a.sync = (uint32_t*)sync_u32; a.err = (uint32_t*)err_u32;
  a.W = (const uint8_t*)qweight; a.s2s = (const uint8_t*)scales_i8; a.s2z = (const uint8_t*)zeros_i8;
  a.wscales = (const half_t*)wscales; a.wsz = (const half_t*)w_szs;
  a.out = (half_t*)out_f16; a.amax = (uint32_t*)amax_slots_u32; a.out_stride = out_row_stride;
  a.M = M; a.N = N; a.K = K; a.clk = (unsigned long long*)clk;
  hipStream_t st = (hipStream_t)stream;
  // (per-channel layers: every producer epilogue is the per-channel one, rows carry a sum; per-group layers: acc * (sw * sa), no sum)
  if ((mode == MODE_CHN) != (sum_f16 != nullptr)) return OMNI_EINVAL;
  if (slab_i32 && (mode == MODE_CHN) != (p_wszs != nullptr)) return OMNI_EINVAL;
  auto with_src = [&](auto src) -> int {
    typedef decltype(src) S;
    return launch_src<S::MODE_OF, S::MODE_OF == MODE_CHN, typename S::Src>(a, src.s, st);
  };
  if (slab_i32) {
    if (mode == MODE_CHN) {
      SrcSlabAddT<true, true> src{(half_t*)residual_f16, (const int32_t*)slab_i32, (size_t)M * K, sk, K, (const half_t*)p_wscales,
                                  (const half_t*)p_wszs, (const half_t*)p_ascales, (const half_t*)p_asums, (half_t)0.0f, (half_t)0.0f};
      return with_src(Tagged<MODE_CHN, SrcSlabAddT<true, true>>{src});
    }
    SrcSlabAddT<false, true> src{(half_t*)residual_f16, (const int32_t*)slab_i32, (size_t)M * K, sk, K, (const half_t*)p_wscales,
                                 (const half_t*)nullptr, (const half_t*)p_ascales, (const half_t*)nullptr, (half_t)0.0f, (half_t)0.0f};
    return with_src(Tagged<MODE_GRP, SrcSlabAddT<false, true>>{src});
  }
  if (delta_f16) {
    const SrcAddB src{(half_t*)residual_f16, (const half_t*)delta_f16, K};
    return mode == MODE_CHN ? with_src(Tagged<MODE_CHN, SrcAddB>{src}) : with_src(Tagged<MODE_GRP, SrcAddB>{src});
  }
  const SrcPlainB src{(const half_t*)residual_f16, K};
  return mode == MODE_CHN ? with_src(Tagged<MODE_CHN, SrcPlainB>{src}) : with_src(Tagged<MODE_GRP, SrcPlainB>{src});
}

extern "C" int omni_w4a8_per_chn_norm_gemm_fused(void* codes_i8, void* residual_f16, const void* slab_i32, int sk,
                                                 const void* delta_f16, const void* p_wscales_f16, const void* p_ascales_f16,
                                                 const void* p_wszs_f16, const void* p_asums_f16, const void* gamma_f16,
                                                 void* sum_f16, void* scale_f16, float eps, const void* qweight,
                                                 const void* wscales_f16, const void* w_szs_f16, void* out_f16,
                                                 long long out_row_stride, void* amax_slots_u32, void* sync_u32, void* err_u32,
                                                 int M, int N, int K, void* clk_u64, void* stream) {
  return norm_gemm_fused(MODE_CHN, codes_i8, residual_f16, slab_i32, sk, delta_f16, p_wscales_f16, p_ascales_f16, p_wszs_f16,
                         p_asums_f16, gamma_f16, sum_f16, scale_f16, eps, qweight, nullptr, nullptr, wscales_f16, w_szs_f16,
                         out_f16, out_row_stride, amax_slots_u32, sync_u32, err_u32, M, N, K, clk_u64, stream);
}

extern "C" int omni_w4a8_per_group_norm_gemm_fused(void* codes_i8, void* residual_f16, const void* slab_i32, int sk,
                                                   const void* delta_f16, const void* p_wscales_f16, const void* p_ascales_f16,
                                                   const void* gamma_f16, void* sum_f16, void* scale_f16, float eps,
                                                   const void* qweight, const void* zeros_i8, const void* scales_i8,
                                                   const void* wscales_f16, void* out_f16, long long out_row_stride,
                                                   void* amax_slots_u32, void* sync_u32, void* err_u32, int M, int N, int K,
                                                   void* clk_u64, void* stream) {
  return norm_gemm_fused(MODE_GRP, codes_i8, residual_f16, slab_i32, sk, delta_f16, p_wscales_f16, p_ascales_f16, nullptr, nullptr,
                         gamma_f16, sum_f16, scale_f16, eps, qweight, zeros_i8, scales_i8, wscales_f16, nullptr, out_f16,
                         out_row_stride, amax_slots_u32, sync_u32, err_u32, M, N, K, clk_u64, stream);
}
