// Host-side tiling / split-K planning for the W4A8 / W8A8 GEMMs (shared by the three GEMM TUs).
#include "qgemm_kernel.h"

namespace omni {

static int g_override_waves = 0;
static int g_override_sk = 0;

GemmPlan plan_gemm(int M, int N, int K, int kalign) {
  GemmPlan pl;
  if (M > 128) {  // MFMA-bound regime: 128 x 256 tile per workgroup, no split
    pl.mb = 8; pl.waves = 4; pl.sk = 1; pl.kslice = K;
    return pl;
  }
  // bandwidth-bound regime (decode): every CU must stream weights.  One wave owns 64 channels
  // and a K-slice; the slice of activations (MT x kslice bytes) must fit the 64 KiB dynamic LDS.
  pl.mb = M <= 16 ? 1 : (M <= 32 ? 2 : (M <= 64 ? 4 : 8));
  const int mt = pl.mb * 16;
  const int ngroups = N / 64;
  pl.waves = ngroups % 4 == 0 ? 4 : (ngroups % 2 == 0 ? 2 : 1);
  auto ok = [&](int s) {
    return s >= 1 && (K % s) == 0 && ((K / s) % kalign) == 0 && mt * (K / s) <= 65536;
  };
  int sk = 1;
  while (!ok(sk) && sk < 1024) ++sk;                // LDS bound first (smallest valid split)
  const int target_waves = 768;                       // ~3 waves per CU, each with 16 KiB in flight
  while (ngroups * sk < target_waves && ok(sk * 2) && (K / (sk * 2)) >= 512) sk *= 2;
  if (!ok(sk)) sk = 1;                                // cannot happen for K % kalign == 0; stay safe
  if (g_override_sk > 0 && ok(g_override_sk)) sk = g_override_sk;
  if (g_override_waves > 0 && g_override_waves <= 4 && ngroups % g_override_waves == 0 &&
      g_override_waves != 3)
    pl.waves = g_override_waves;
  pl.sk = sk;
  pl.kslice = K / sk;
  return pl;
}

}  // namespace omni

extern "C" void omni_gemm_set_plan_override(int waves, int sk) {
  omni::g_override_waves = waves;
  omni::g_override_sk = sk;
}

extern "C" void omni_gemm_get_plan(int M, int N, int K, int kalign, int* mb, int* waves, int* sk) {
  omni::GemmPlan pl = omni::plan_gemm(M, N, K, kalign);
  if (mb) *mb = pl.mb;
  if (waves) *waves = pl.waves;
  if (sk) *sk = pl.sk;
}

extern "C" size_t omni_gemm_workspace_bytes(int M, int N, int K) {
  if (M < 1 || N < 64 || K < 64) return 0;
  // upper bound over the three GEMM flavours (per-group needs 128-aligned slices, so never more splits)
  omni::GemmPlan pl = omni::plan_gemm(M, N, K, 64);
  return pl.sk > 1 ? (size_t)pl.sk * M * N * sizeof(int32_t) : 0;
}

extern "C" int omni_abi_version(void) { return 1; }
