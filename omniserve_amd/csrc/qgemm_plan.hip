// Host-side tiling / split-K planning for the W4A8 / W8A8 GEMMs (shared by the three GEMM TUs).
#include "qgemm_kernel.h"

namespace omni {

static int g_override_waves = 0;
static int g_override_sk = 0;

GemmPlan plan_gemm(int M, int N, int K, int kalign, bool deferred) {
  GemmPlan pl;
  pl.kw = 1;
  if (M > 128) {  // MFMA-bound regime: 128 x 256 tile per workgroup, no split
    pl.mb = 8; pl.waves = 4; pl.sk = 1; pl.kslice = K;
    return pl;
  }
  // bandwidth-bound regime (decode): every CU must stream weights.  One wave owns 64 channels
  // and a K-slice; activations are staged per round of RING steps, so the slice length is free.
  pl.mb = M <= 16 ? 1 : (M <= 32 ? 2 : (M <= 64 ? 4 : 8));
  pl.waves = pl.mb == 8 ? 4 : 1;   // fixed per tile height (GemvCfg)
  const int ngroups = N / 64;
  const int round_k = (pl.mb <= 2 ? 8 : 4) * 64;     // k per ring round (W8A8 uses 4 x 64 <= this)
  auto ok = [&](int s) { return s >= 1 && (K % s) == 0 && ((K / s) % kalign) == 0; };
  int sk = 1;
  if (pl.mb == 1) {
    // M <= 16: single-wave tiles, K split first inside the workgroup (kw waves, no slab traffic) and only
    // then across workgroups.  Measured on MI355X (tools/kw_sweep*.py): a wave should stream <= 1024 k
    // (two ring rounds) when the split is free (one kernel), <= 512 k when the int32 slabs are consumed
    // by a fused kernel anyway (deferred) or a split cannot be avoided.
    int kw = 4;
    while (kw > 1 && (K % (kw * kalign)) != 0) kw >>= 1;
    auto part = [&](int s) { return K / (s * kw); };
    auto fits = [&](int s) { return ok(s) && ((K / s) % (kw * kalign)) == 0; };
    if (!deferred && part(1) <= 1024) {
      sk = 1;
    } else {
      int best = 0;
      for (int s = 1; s <= 64; ++s) {
        if (!fits(s) || part(s) < 256) continue;
        if (part(s) <= 512) { best = s; break; }
        best = s;   // largest usable split so far
      }
      sk = best > 0 ? best : 1;
    }
    if (g_override_sk > 0 && ok(g_override_sk)) sk = g_override_sk;
    if (g_override_waves == 1 || g_override_waves == 2 || g_override_waves == 4) kw = g_override_waves;
    while (kw > 1 && ((K / sk) % (kw * kalign)) != 0) kw >>= 1;
    pl.kw = kw;
  } else if (pl.mb <= 4) {
    // 16 < M <= 64: single-wave tiles of MB*16 rows, two K parts per workgroup; split across workgroups until a
    // wave streams <= 2048 k (a wave keeps only RING x 2 KiB in flight, so short parts = more bytes in flight)
    pl.waves = 1;
    int kw = 2;
    while (kw > 1 && (K % (kw * kalign)) != 0) kw >>= 1;
    auto fits = [&](int s) { return ok(s) && ((K / s) % (kw * kalign)) == 0; };
    int best = 1;
    for (int s = 1; s <= 64; ++s) {
      if (!fits(s) || K / (s * kw) < 512) continue;
      best = s;
      if (K / (s * kw) <= 2048) break;
    }
    sk = best;
    if (g_override_sk > 0 && ok(g_override_sk)) sk = g_override_sk;
    if (g_override_waves == 1 || g_override_waves == 2) kw = g_override_waves;
    while (kw > 1 && ((K / sk) % (kw * kalign)) != 0) kw >>= 1;
    pl.kw = kw;
  } else {
    auto full_rounds = [&](int s) { return ((K / s) % round_k) == 0; };
    // M <= 128: four channel groups per workgroup share the staged activations; split K until ~2 waves per CU
    const int target_waves = 448;
    for (int s = 1; s <= 64 && ngroups * sk < target_waves; ++s) {
      if (!ok(s) || (K / s) < round_k) continue;
      if (!full_rounds(s) && full_rounds(sk)) continue;
      sk = s;
    }
    if (g_override_sk > 0 && ok(g_override_sk)) sk = g_override_sk;
  }
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
  omni::GemmPlan pl = omni::plan_gemm(M, N, K, kalign, false);
  if (mb) *mb = pl.mb;
  if (waves) *waves = pl.waves;
  if (sk) *sk = pl.sk;
}

extern "C" size_t omni_gemm_workspace_bytes(int M, int N, int K) {
  if (M < 1 || N < 64 || K < 64 || M > 128) return 0;   // M > 128: no split, no scratch
  // upper bound over the three GEMM flavours (per-group needs 128-aligned slices, so never more splits)
  // the deferred (slab-only) plan never splits less than the plain one
  omni::GemmPlan pl = omni::plan_gemm(M, N, K, 64, true);
  return (size_t)pl.sk * M * N * sizeof(int32_t);
}

extern "C" int omni_abi_version(void) { return 1; }
