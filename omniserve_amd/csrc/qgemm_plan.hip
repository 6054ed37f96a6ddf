// Host-side tiling / split-K planning for the W4A8 / W8A8 GEMMs (shared by the three GEMM TUs).
#include "qgemm_kernel.h"

namespace omni {

// (plan overrides: test / sweep hooks, per enqueueing thread like everything else here)
static thread_local int g_override_waves = 0;
static thread_local int g_override_sk = 0;
static thread_local int g_midm_mode = -1;     // omni_gemm_set_midm_override
static thread_local int g_midm_sk = 0;
static thread_local PrefetchArgs g_armed_prefetch = {};   // per enqueueing thread: armed and consumed by the same caller
// weight tensors named by omni_prefetch_arm_gemm whose GEMV has not been enqueued yet (one-shot, per thread): that GEMV
// reads its weights with plain loads (they sit in L2), every other one streams non-temporally.  A decode layer has at
// most two such entries alive at a time (down_proj armed while gate_up runs); 4 slots, oldest replaced.
static thread_local const void* g_prefetched_w[4] = {nullptr, nullptr, nullptr, nullptr};
static thread_local unsigned g_prefetched_next = 0;

bool take_prefetched_weight(const void* weight) {
  if (!weight) return false;
  for (auto& w : g_prefetched_w)
    if (w == weight) { w = nullptr; return true; }
  return false;
}
static void note_prefetched_weight(const void* weight) {
  for (auto& w : g_prefetched_w)
    if (w == weight) return;
  g_prefetched_w[g_prefetched_next++ & 3] = weight;
}

PrefetchArgs take_armed_prefetch() {
  PrefetchArgs pf = g_armed_prefetch;
  g_armed_prefetch = PrefetchArgs{};
  return pf;
}

GemmPlan plan_gemm(int M, int N, int K, int kalign, bool deferred, bool w8) {
  GemmPlan pl;
  pl.kw = 1;
  pl.mz = 1;
  pl.narrow = 0;
  pl.midm = 0;
  // Mid-M kernel (qgemm_midm.h): M = 33 .. 128 rows (one row tile), N in whole 128-channel tiles, K slices of whole 256-k
  // chunks.  A launch of it carries ~7 us that do not depend on K (first requests, the K-phase exchange, stores) and streams
  // at about twice the single-wave tiles' rate beyond them, so it takes the BIG matrices (measured, tools/midm_sweep.py,
  // profiles/r05_a: Llama-3-8B gate_up at M = 128 35.4 -> 23.1 us, Llama-2-70B gate_up 138 -> 71, down 106 -> 41 -- and
  // Llama-3-8B qkv 11.7 -> 19.6, which stays where it was): N x K >= 80 M weights, or >= 50 M where the caller wants the
  // slabs anyway (the deferred forms: no slab epilogue launch to pay for; Llama-3-8B down at M = 64: 11.4 -> 10.7 us, g128
  // 13.6 -> 12.6; g128 gate_up at M = 64 21.2 -> 19.5).
  // K split: one workgroup per CU and launch (its LDS rings fill the CU) -- the largest split that keeps the grid inside one
  // round of 256 and >= 4 chunks per slice (Llama-2-70B qkv, 80 tiles: 2 slices 28.5 us, 4 slices 34.7).
  // (the kernel addresses weights and activations with 32-bit lane offsets: matrices of 4 GiB and more keep the other tiles)
  const bool midm_fits = (unsigned long long)N * K * (w8 ? 2 : 1) < (2ull << 32) && (unsigned long long)M * K < (1ull << 32);
  if (g_midm_mode != 0 && g_override_waves == 0 && g_override_sk == 0 && M > 32 && M <= 128 && N % 128 == 0 && K % KCHUNK == 0 &&
      midm_fits) {
    const long long weights = (long long)N * K;
    bool take = g_midm_mode > 0;
    if (g_midm_mode < 0) take = weights >= 80000000LL || (deferred && weights >= 50000000LL);
    if (take) {
      const int tiles = N / 128;
      int best = 1;
      for (int s2 = 2; s2 <= 16; ++s2) {
        if (K % (s2 * KCHUNK) != 0 || (K / s2) % kalign != 0 || K / s2 < 4 * KCHUNK || tiles * s2 > 256) continue;
        best = s2;
      }
      if (g_midm_sk > 0 && K % (g_midm_sk * KCHUNK) == 0) best = g_midm_sk;
      pl.midm = 1; pl.mb = M <= 64 ? 4 : 8; pl.mz = 1; pl.waves = 2; pl.kw = 1;
      pl.sk = best; pl.kslice = K / best;
      return pl;
    }
  }
  if (M > 128) {  // MFMA-bound regime: 128 x 256 tile per workgroup
    pl.mb = 8; pl.waves = 4; pl.sk = 1; pl.kslice = K;
    // Few tiles (decode at batch 129..512: the published A100 figure is quoted at bs = 256): Llama-3-8B's down_proj is
    // 2 x 16 = 32 tiles of 56 K chunks on 256 CUs (80 us per layer at bs = 256, profiles/r04_b), o_proj 32, qkv 48.  Split K over
    // grid.y into int32 slabs (+ the slab epilogue launch) until about one workgroup per CU streams: the smallest split with
    // >= 256 workgroups, whole 256-k chunks and >= 1024 k per workgroup.  A grid that already covers half the chip stays whole.
    const long long tiles = (long long)((M + 127) / 128) * ((N + 255) / 256);
    if (M <= 512 && tiles <= 128 && g_override_waves == 0) {
      int best = 1;
      for (int s2 = 2; s2 <= 8; s2 *= 2) {
        if (K % (s2 * KCHUNK) != 0 || K / s2 < 1024 || tiles * s2 > 512) break;
        best = s2;
        if (tiles * s2 >= 256) break;
      }
      if (g_override_sk > 0 && K % (g_override_sk * KCHUNK) == 0) best = g_override_sk;
      pl.sk = best;
      pl.kslice = K / best;
    }
    return pl;
  }
  // bandwidth-bound regime (decode): every CU must stream weights.  One wave owns 64 channels
  // and a K-slice; activations are staged per round of RING steps, so the slice length is free.
  pl.mb = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
  pl.mz = M <= 64 ? 1 : 2;         // M = 65..128: two 64-row tiles per channel group
  pl.waves = 1;                    // fixed per tile height (GemvCfg)
  const int ngroups = N / 64;
  auto ok = [&](int s) { return s >= 1 && (K % s) == 0 && ((K / s) % kalign) == 0; };
  int sk = 1;
  // M = 33..128, int4 modes: 32-row tiles (ceil(M / 32) row tiles per channel group, the re-reads of a weight byte are
  // served by L1 / L2) beat the 64-row tile whenever their grid still fits the chip in one round -- smaller serial
  // chains per wave at the same two waves per SIMD (measured, profiles/r03_c_*: Llama-2-70B TP=8 shard at M = 128
  // gate_up 24.8 -> 21.6 us, down 18.4 -> 12.9, qkv 12.5 -> 10.0, o 7.6 -> 5.6; Llama-3-8B g128 at M = 64 down 20.4 ->
  // 18.6, qkv 12.8 -> 10.1, o 10.9 -> 9.9, but gate_up 23.0 -> 31.5: 896 workgroups, two rounds -- stays on 64 rows).
  static const int narrow_mode = omni_knob("OMNI_GEMV_NARROW", 1);   // 0: never (A/B, tuning builds)
  // (g128 at M <= 64: faster launch by launch, but the Llama-3-8B bs = 64 decode step measured 1.2 % slower with it -- off)
  if (!w8 && M > 32 && !(kalign == 128 && M <= 64) && narrow_mode != 0 && g_override_waves == 0 && g_override_sk == 0) {
    const int mz = (M + 31) / 32;
    if (ngroups * mz <= 640 && (K % (4 * kalign)) == 0) {
      pl.mb = 2; pl.mz = mz; pl.waves = 1; pl.narrow = 1; pl.kw = 4;
      // smallest split with <= 2048 k per wave and >= 128 workgroups; parts of at least 256 k
      int best = 1;
      for (int s = 1; s <= 64; ++s) {
        if (!ok(s) || ((K / s) % (4 * kalign)) != 0 || K / (s * 4) < 256) continue;
        best = s;
        if (K / (s * 4) <= 2048 && ngroups * mz * s >= 128) break;
      }
      pl.sk = best;
      pl.kslice = K / best;
      return pl;
    }
  }
  if (pl.mb == 1) {
    // M <= 16: single-wave tiles, K split first inside the workgroup (kw waves, no slab traffic) and only
    // then across workgroups.  Measured on MI355X (tools/kw_sweep*.py): a wave should stream <= 1024 k
    // (two ring rounds) when the split is free (one kernel), <= 512 k when the int32 slabs are consumed
    // by a fused kernel anyway (deferred) or a split cannot be avoided.
    int kw = 4;
    while (kw > 1 && (K % (kw * kalign)) != 0) kw >>= 1;
    auto part = [&](int s) { return K / (s * kw); };
    auto fits = [&](int s) { return ok(s) && ((K / s) % (kw * kalign)) == 0; };
    // W8A8 rows are twice the bytes per k: the deferred (slab-only) plan halves the k per wave (same bytes per wave)
    static const int part_w4_deferred = omni_knob("OMNI_DEFERRED_PART", 512);   // (A/B knob, tuning builds)
    const int part_target = (deferred && w8) ? 256 : (deferred ? part_w4_deferred : 512);
    // W8A8 rows are twice the bytes per k: with few channel groups (qkv at batch 1: 96 workgroups streaming 256 KiB each)
    // a split to 512 k per wave + the slab epilogue launch beats the single kernel (OMNI_W8_SMALL_SPLIT=0: off, A/B)
    static const int w8_small_split = omni_knob("OMNI_W8_SMALL_SPLIT", 1);
    const bool w8_small = w8 && !deferred && w8_small_split && ngroups <= 128 && part(1) > 512;
    if (!deferred && part(1) <= 1024 && !w8_small) {
      sk = 1;
    } else {
      int best = 0;
      for (int s = 1; s <= 64; ++s) {
        if (!fits(s) || part(s) < 256) continue;
        if (part(s) <= part_target) { best = s; break; }
        best = s;   // largest usable split so far
      }
      sk = best > 0 ? best : 1;
    }
    if (g_override_sk > 0 && ok(g_override_sk)) sk = g_override_sk;
    if (g_override_waves == 1 || g_override_waves == 2 || g_override_waves == 4) kw = g_override_waves;
    while (kw > 1 && ((K / sk) % (kw * kalign)) != 0) kw >>= 1;
    pl.kw = kw;
  } else {
    // 16 < M <= 128: single-wave tiles of MB*16 rows, two K parts per workgroup; split across workgroups until a
    // wave streams <= 2048 k (a wave keeps only RING x 2 KiB in flight, so short parts = more bytes in flight).
    // M > 32 (64-row tiles: 20-us kernels, the 5-us slab epilogue is cheap next to them) keeps splitting, down to
    // 512 k per wave, until ~1.5 workgroups per CU are in flight: at bs = 64 the Llama-3-8B qkv / o projections ran
    // on 96 / 64 workgroups at 0.6 TB/s (profiles/r02_*).
    int kw = (pl.mb == 4 && !w8 && OMNI_GEMV_AR_MB4 == 2) ? 4 : 2;
    while (kw > 1 && (K % (kw * kalign)) != 0) kw >>= 1;
    auto fits = [&](int s) { return ok(s) && ((K / s) % (kw * kalign)) == 0; };
    // a part should be whole ring rounds (the remainder runs one unpipelined step at a time)
    const int ring_k = 64 * (w8 ? 4 : (pl.mb == 4 ? OMNI_GEMV_RING_MB4 : 8));
    int best = 0;
    for (int strict = 1; strict >= 0 && best == 0; --strict) {
      for (int s = 1; s <= 64; ++s) {
        const int part = K / (s * kw);
        if (!fits(s) || part < (kw == 4 ? 256 : 512) || (strict && (part % ring_k) != 0)) continue;
        best = s;
        if (kw == 4) {   // 64-row tile, four K parts per workgroup: <= 1024 k per wave and >= 768 waves (measured, r02_e)
          if (part <= 1024 && ngroups * s * pl.mz * kw >= 768) break;
        } else if (part <= 2048 && (pl.mb < 4 || ngroups * s * pl.mz >= 384)) {
          break;
        }
      }
    }
    if (best == 0) best = 1;
    sk = best;
    if (g_override_sk > 0 && ok(g_override_sk)) sk = g_override_sk;
    if (g_override_waves == 1 || g_override_waves == 2 || (g_override_waves == 4 && kw == 4)) kw = g_override_waves;
    while (kw > 1 && ((K / sk) % (kw * kalign)) != 0) kw >>= 1;
    pl.kw = kw;
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

extern "C" void omni_gemm_set_midm_override(int mode, int sk) {
  omni::g_midm_mode = mode;
  omni::g_midm_sk = sk;
}

extern "C" void omni_gemm_get_plan(int M, int N, int K, int kalign, int* mb, int* waves, int* sk) {
  omni::GemmPlan pl = omni::plan_gemm(M, N, K, kalign, false, false);
  if (mb) *mb = pl.mb;
  if (waves) *waves = pl.waves;
  if (sk) *sk = pl.sk;
}

// Would the row-kernel-free decode forms (omni_*_gemm_silu for gate_up, omni_*_gemm_partial_f16 for o / down) accept a layer
// with these dimensions?  The same plan conditions launch_gemm_silu / launch_gemm_partial_f16 check at launch (one place to
// keep in step: qgemm_kernel.h), so that a driver can choose its fusion level BEFORE its first step instead of running into
// EINVAL there (hidden > 4096: the gate_up plan splits K across workgroups; rows > N / 64: no rider per row).
// mode: 0 per-channel W4A8 (row sums needed), 1 per-group W4A8, 2 W8A8.
extern "C" int omni_gemm_rowfree_ok(int M, int hidden, int attn_dim, int inter, int mode) {
  using namespace omni;
  if (mode < 0 || mode > 2 || M < 1 || M > 16 || hidden < 64 || attn_dim < 64 || inter < 64) return 0;
  const int kalign = mode == MODE_GRP ? 128 : 64;
  const bool w8 = mode == MODE_W8;
  if (hidden % 64 || attn_dim % kalign || inter % kalign || (2 * inter) % 128 || hidden % kalign) return 0;
  const GemmPlan gu = plan_gemm(M, 2 * inter, hidden, kalign, false, w8);
  if (gu.sk != 1 || gu.mb != 1) return 0;
  const int ar = w8 ? GemvCfg<1, MODE_W8>::AR : GemvCfg<1, MODE_CHN>::AR;
  for (const int K : {attn_dim, inter}) {
    if (hidden / 64 < M) return 0;
    const GemmPlan pl = plan_gemm(M, hidden, K, kalign, true, w8);
    if (pl.mb != 1 || pl.kw < 2) return 0;
    if (mode == MODE_CHN && (size_t)K * sizeof(float) > (size_t)pl.kw * 2 * 16 * ar * KSTEP) return 0;   // rider's LDS row
  }
  return 1;
}

// Largest K split any plan of an (M, N, K) problem can take: the maximum over the K alignments of the three GEMM flavours (64;
// 128 for per-group), over W4 / W8 rows and over the mid-M kernel off / forced on WITH every legal forced split (so that a
// buffer sized once stays large enough under omni_gemm_set_midm_override(1, sk), whatever the mode was when it was sized;
// the loop used to skip forced mode 1 and the forced sk).  `deferred`: the slab-only forms' plans.
static int max_plan_sk(int M, int N, int K, bool deferred) {
  using namespace omni;
  int sk = 1;
  const int keep_mode = g_midm_mode, keep_sk = g_midm_sk;
  for (int midm = 0; midm < 2; ++midm) {
    g_midm_mode = midm;
    for (int force = midm ? 16 : 0; force >= 0; force = (force > 2 ? force - 1 : (force == 2 ? 0 : -1))) {
      g_midm_sk = force;      // 0: the heuristic split; 2 .. 16: omni_gemm_set_midm_override(1, force)
      for (int kalign = 64; kalign <= 128; kalign *= 2) {
        if (K % kalign != 0) continue;
        for (int w8 = 0; w8 < 2; ++w8) {
          const GemmPlan pl = plan_gemm(M, N, K, kalign, deferred, w8 != 0);
          if (pl.sk > sk) sk = pl.sk;
        }
      }
    }
  }
  g_midm_mode = keep_mode;
  g_midm_sk = keep_sk;
  return sk;
}

// plain entry points (omni_*_gemm): scratch only where a plan splits K (0 otherwise -- ADVICE r5: M = 129 .. 512 used to pin one
// M x N slab here for the sake of the slab-only forms, which now have their own query)
extern "C" size_t omni_gemm_workspace_bytes(int M, int N, int K) {
  if (M < 1 || N < 64 || K < 64) return 0;
  const int sk = max_plan_sk(M, N, K, false);
  return sk > 1 ? (size_t)sk * M * N * sizeof(int32_t) : 0;
}

// slab-only forms (omni_*_gemm_partial[_f16], M <= 512): they write their int32 accumulators even when the plan keeps K whole
extern "C" size_t omni_gemm_partial_workspace_bytes(int M, int N, int K) {
  if (M < 1 || M > 512 || N < 64 || K < 64) return 0;
  const int a = max_plan_sk(M, N, K, true), b = max_plan_sk(M, N, K, false);
  return (size_t)(a > b ? a : b) * M * N * sizeof(int32_t);
}

// ---- fused extension: L2 weight prefetch riding on the next row kernel (common.h: PrefetchArgs) -------------------
extern "C" int omni_prefetch_arm_gemm(const void* weight, int M, int N, int K, int mode, int deferred,
                                      int64_t budget_bytes, int blocks) {
  using namespace omni;
  g_armed_prefetch = PrefetchArgs{};
  if (!weight || blocks <= 0 || budget_bytes <= 0) {                        // disarm (and forget every pending load-policy entry)
    if (!weight)
      for (auto& w : g_prefetched_w) w = nullptr;
    return OMNI_OK;
  }
  // mode | 0x10: the gate_up form with the fused SiLU epilogue (omni_w4a8_per_*_gemm_silu): workgroup x streams tile row x
  // of the gate half and tile row N/64 + x of the up half
  const bool silu = (mode & 0x10) != 0;
  // mode | 0x20: the consuming GEMV keeps its non-temporal weight loads (the prefetch then only warms the memory-side
  // cache); default: that GEMV -- the next decode-shape GEMM enqueued by this thread on this weight tensor -- loads plain
  const bool keep_nt = (mode & 0x20) != 0;
  mode &= 0xF;
  if (silu && (mode == MODE_W8 || deferred || (N / 64) % 8 != 0)) return mode == MODE_W8 ? OMNI_EINVAL : OMNI_OK;
  if (mode < 0 || mode > 2 || M < 1 || M > 128 || N % 64 != 0 || K % 64 != 0 || K < 64) return OMNI_EINVAL;
  const GemmPlan pl = plan_gemm(M, N, K, mode == MODE_GRP ? 128 : 64, deferred != 0, mode == MODE_W8);
  PrefetchArgs pf{};
  pf.base = static_cast<const uint8_t*>(weight);
  const int groups_per_wg = pl.waves;                                       // 64-channel groups per workgroup
  pf.gx = (N / 64 + groups_per_wg - 1) / groups_per_wg;
  pf.gy = pl.sk;
  pf.kw = pl.kw;
  if ((N / 64) % groups_per_wg != 0) return OMNI_OK;                         // ragged last workgroup: skip the hint
  if (mode == MODE_W8) {
    // [N][K] int8 rows of K bytes: a wave walks its 64 rows together along k, so a budget below the matrix size
    // fetches whole rows of the first rows of every workgroup (K-parts of a row can be shorter than a 1-KiB piece)
    if (pf.gy > 1 && (pf.gx & 7) != 0) return OMNI_OK;                       // K-slices of a row group on different XCDs
    pf.row_bytes = K; pf.rows_per_wg = 64 * groups_per_wg;
    pf.gy = 1; pf.kw = 1;
    const long long total = (long long)N * K;
    double f = total > budget_bytes ? (double)budget_bytes / (double)total : 1.0;
    pf.rows_pf = (int)(pf.rows_per_wg * f);
    pf.pf_bytes = (K / 1024) * 1024;
    if (pf.rows_pf < 1 || pf.pf_bytes < 1024) return OMNI_OK;
  } else {
    pf.row_bytes = (long long)K * 16; pf.rows_per_wg = 2 * groups_per_wg;
    if (silu) {
      // as 2 * (N/64) single-row workgroups: item v = tile row v is fetched on XCD v % 8, and (N/64) % 8 == 0 makes that
      // the XCD of the real workgroup v % (N/64) for both of its rows
      pf.rows_per_wg = 1;
      pf.gx = N / 32;
    }
    pf.rows_pf = pf.rows_per_wg;
    const long long part_bytes = pf.row_bytes / ((long long)pf.gy * pf.kw);
    const long long total = pf.row_bytes * pf.rows_per_wg * pf.gx;
    long long want = part_bytes;
    if (total > budget_bytes) want = (long long)((double)part_bytes * (double)budget_bytes / (double)total);
    want = (want / 1024) * 1024;
    if (want < 1024 || part_bytes < 1024) return OMNI_OK;                    // parts too small for 1-KiB pieces: skip
    pf.pf_bytes = (int)want;
  }
  pf.blocks = (blocks + 7) & ~7;
  static const int delay = omni_knob("OMNI_PREFETCH_DELAY", 0);
  pf.delay = delay;
  g_armed_prefetch = pf;
  if (!keep_nt) note_prefetched_weight(weight);     // the GEMV on this weight tensor reads L2-resident lines: plain loads
  return OMNI_OK;
}

// 3: omni_gemm_set_weight_policy removed (the load policy is per call: omni_prefetch_arm_gemm mode bit 0x20);
//    omni_gemm_workspace_bytes covers the slab-only forms (>= one slab up to 512 rows); omni_gemm_set_midm_override added.
// 4: the slab-only forms have their own query (omni_gemm_partial_workspace_bytes) and omni_gemm_workspace_bytes is the plain
//    entry points' again (0 without a K split); both cover every legal omni_gemm_set_midm_override; the (norm -> GEMV) pair
//    entry points (omni_w4a8_per_*_norm_gemm_fused, omni_norm_gemm_fused_ok) added.
extern "C" int omni_abi_version(void) { return 4; }
