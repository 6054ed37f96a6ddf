// Varlen causal prefill attention (dense heads and token-streaming "Lambda" heads) for MI355X (gfx950).
//
// Replaces the un-vendored third-party kernels the reference calls for ALL prefill attention
// (omniserve/modeling/layers/ctx_attn/ctx_attn_func.py:39-45,68-73):
//   block_sparse_attn.flash_attn_varlen_func(q,k,v,cu_q,cu_k,max_q,max_k,dropout_p=0,causal=True)
//   block_sparse_attn.token_streaming_attn_func(q,k,v,cu_q,cu_k,head_mask_type,streaming_info,max_q,max_k)
// Semantics (inferred, the package is not vendored): softmax(q k^T / sqrt(Dh) + mask) v per
// sequence and q head, GQA by head index division; mask = causal (bottom-right aligned) and, for heads with
// head_mask_type < 0, additionally (k_pos < sink  OR  q_pos - k_pos < local).
//
// Structure (flash-attention, online softmax, matrix cores): a workgroup of 8 waves owns 128 query rows (16 per
// wave = one 16-row MFMA block; ~120 VGPRs per wave, so two workgroups = 4 waves per SIMD cover the softmax VALU
// work of one another) of one q head and walks the keys 64 at a time.  The K and V tiles are staged once per
// workgroup in LDS, double buffered: the next tile is issued as LDS-DMA pieces (global_load_lds, 1 KiB per wave
// instruction, no VGPR staging) into the other buffer before the MFMAs of the current one, and the single barrier
// per tile carries their vmcnt(0); a key row fetched from L2 serves 128 queries.  S^T = K Q^T (A = K rows from
// LDS, XOR-swizzled 16-B slots -- applied on the global side of the DMA -- : conflict-free ds_read_b128; B = Q
// held in registers) leaves a lane with 4+4 keys of ONE query row per 32 keys -- exactly the B-operand layout of
// the second product O^T = V^T P^T, whose A operand (V transposed) is read from the V tile with
// ds_read_b64_tr_b16.  Probabilities never leave registers; row statistics reduce over the 4 lanes that share a
// query row.  Tiles that need no masking skip the per-element predicate (workgroup-uniform).  With q_tiles > 0 the
// grid is 1-D and XCD-aware (prefill_map_block: the q heads of a kv head share an XCD's L2; dense work balanced over XCDs).
#include "common.h"

namespace omni {

constexpr int PDH = 128;
constexpr int PKT = 64;               // keys per tile
#ifndef OMNI_PREFILL_PQB
#define OMNI_PREFILL_PQB 1
#endif
constexpr int PQB = OMNI_PREFILL_PQB;   // 16-row query blocks per wave
#ifndef OMNI_PREFILL_MFMA32
#define OMNI_PREFILL_MFMA32 0          // default kernel: 0 = 16-row form, 1 = 32-row form (prefill_attn32_kernel); omni_prefill_set_variant overrides
#endif
#ifndef OMNI_PREFILL_ABLATE_DMA
#define OMNI_PREFILL_ABLATE_DMA 0     // timing experiment (wrong results): no tile DMA after the first tile
#endif

#ifndef OMNI_PREFILL_WAVES
#define OMNI_PREFILL_WAVES (8 / OMNI_PREFILL_PQB)
#endif
constexpr int PWAVES = OMNI_PREFILL_WAVES;         // 8 waves x 16 rows: ~120 VGPRs per wave, 4 waves per SIMD hide the softmax VALU work
constexpr int PPT = (PKT * 16) / (64 * PWAVES);   // 1-KiB LDS-DMA pieces of a K (or V) tile per wave
constexpr int PQROWS = 16 * PQB * PWAVES;   // 128 query rows per workgroup
constexpr int PKROW = 256;            // bytes per key row of the K tile (swizzled slots)
constexpr int PVROW = 256;            // bytes per key row of the V tile (swizzled slots, see the kernel)
constexpr int PKTILE = PKT * PKROW, PVTILE = PKT * PVROW;
typedef __fp16 pv4hp __attribute__((__vector_size__(4 * sizeof(__fp16))));

__device__ __forceinline__ int xor_nohoist(int a, int uniform_b) {
  int r;
  asm volatile("v_xor_b32 %0, %2, %1" : "=v"(r) : "v"(a), "s"(uniform_b));
  return r;
}

struct PrefillArgs {
  const half_t* q; const half_t* k; const half_t* v; half_t* out;
  int64_t q_stride, k_stride, v_stride;     // elements between consecutive tokens
  const int* cu_q; const int* cu_k;
  const int* head_mask_type;                 // [Hq] or null (all dense)
  const int* streaming_info;                 // [2*Hq] (sink, local) or null
  int num_heads, num_kv_heads;
  int causal;
  int q_tiles;                               // > 0: 1-D grid, XCD-aware order (prefill_map_block)
  int xcd_split;                             // 1-D grid: XCDs that share the query tiles of one kv head (1, 2, 4 or 8)
  int gran_log2;                             // granularity of (sink, local): 0 tokens (token_streaming_attn_func), 7 blocks of 128
};                                           // tokens (block_streaming_attn_func): see first_local_key below

// Streaming heads see the keys  key < sink << g  and  key >= first_local_key(qpos): the first key of the local window of the
// query at (bottom-right aligned) position qpos.  g = 0: qpos - local + 1 (the last `local` tokens, itself included); g = 7:
// the first token of block (qpos >> 7) - local + 1 (the query's own 128-token block and the local - 1 blocks before it).
__device__ __forceinline__ int first_local_key(int qpos, int local, int g) { return ((qpos >> g) - local + 1) * (1 << g); }

// 1-D grid: workgroups are dealt round-robin to the 8 XCDs (private L2s; the dispatch is static: workgroup w runs on XCD
// w % 8 whatever the others are doing).  All q heads of a kv head share an XCD's L2, so a K/V tile is fetched from the
// fabric once per XCD that works on the head; the query tiles of a kv head are spread over W = xcd_split XCDs (tile t of
// part t % W) so that every XCD gets the same amount of DENSE work when some kv heads are streaming heads (LServe: a dense
// head costs O(L^2), a streaming head O(L * window); with whole kv heads pinned to XCDs the XCDs that drew streaming heads
// idle for most of the launch).  kv heads are taken dense-first: sorted position pos, part j -> XCD (pos * W + j) % 8.
// Returns false for the padding workgroups of a part that has fewer tiles.
__device__ __forceinline__ bool prefill_map_block(const PrefillArgs& p, int& b, int& h, int& qt) {
  const int W = p.xcd_split, Hk = p.num_kv_heads, G = p.num_heads / Hk;
  const int P = (Hk * W) >> 3;                       // (kv head, part) pairs per XCD
  const int R = (p.q_tiles + W - 1) / W;             // tile rows of a part
  const int wid = blockIdx.x, xcd = wid & 7;
  int slot = wid >> 3;
  const int u = slot % G; slot /= G;
  const int i = slot % P; slot /= P;
  const int r = slot % R;
  b = slot / R;
  const int idx = xcd + 8 * i, pos = idx / W, j = idx - pos * W;
  qt = p.q_tiles - 1 - (r * W + j);                  // long (late) query tiles first
  int kvh = pos;
  if (p.head_mask_type != nullptr) {                 // pos-th kv head in dense-first order (Hk scalar loads, once)
    int nd = 0;
    for (int t = 0; t < Hk; ++t) nd += p.head_mask_type[t * G] >= 0;
    const bool want_dense = pos < nd;
    int left = want_dense ? pos : pos - nd;
    kvh = 0;
    for (int t = 0; t < Hk; ++t) {
      const bool dense = p.head_mask_type[t * G] >= 0;
      if (dense == want_dense) {
        if (left == 0) { kvh = t; break; }
        --left;
      }
    }
  }
  h = kvh * G + u;
  return qt >= 0;
}

#ifndef OMNI_PREFILL_MIN_BLOCKS
#define OMNI_PREFILL_MIN_BLOCKS 2
#endif
__global__ __launch_bounds__(64 * PWAVES, OMNI_PREFILL_MIN_BLOCKS) __attribute__((amdgpu_waves_per_eu(4 / PQB, 4 / PQB), amdgpu_num_vgpr(512 / (4 / PQB))))   // 128 VGPRs (PQB = 1) or 256 (PQB = 2)
void prefill_attn_kernel(PrefillArgs p) {
  // four separate LDS objects and a loop body instantiated per buffer parity (static indices): with one array and a
  // runtime buffer index the compiler cannot tell the DMA target from the tile being read and puts vmcnt(0) -- the
  // whole flight time of the next tile -- in front of the first LDS read of every iteration
  __shared__ __attribute__((aligned(16))) uint8_t ktile0[PKTILE];
  __shared__ __attribute__((aligned(16))) uint8_t ktile1[PKTILE];
  __shared__ __attribute__((aligned(16))) uint8_t vtile0[PVTILE];
  __shared__ __attribute__((aligned(16))) uint8_t vtile1[PVTILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  int b = blockIdx.z, h = blockIdx.y, qt = gridDim.x - 1 - blockIdx.x;   // long (late) query tiles are scheduled first
  if (p.q_tiles > 0 && !prefill_map_block(p, b, h, qt)) return;
  const int hk = h / (p.num_heads / p.num_kv_heads);
  const int q_begin = p.cu_q[b], len_q = p.cu_q[b + 1] - q_begin;
  const int k_begin = p.cu_k[b], len_k = p.cu_k[b + 1] - k_begin;
  const int q_first = qt * PQROWS;
  if (q_first >= len_q) return;                                 // whole workgroup out of range
  const int q_last = min(q_first + PQROWS, len_q) - 1;
  const int q0 = q_first + wave * (16 * PQB);                   // first query row of this wave
  const int off = len_k - len_q;                                // bottom-right aligned causal mask
  const bool streaming = p.head_mask_type != nullptr && p.head_mask_type[h] < 0;
  const int gl = p.gran_log2;
  const int sink = streaming ? p.streaming_info[2 * h] << gl : 0;      // in tokens
  const int local = streaming ? p.streaming_info[2 * h + 1] : 0;       // in units of 1 << gl tokens
  const float scale2 = 0.08838834764831845f * 1.4426950408889634f;   // log2(e)/sqrt(128): scores in the exp2 domain

  // B operand of S^T = K Q^T: Q[row][32s + 8*l4 + (0..7)] for the wave's two row blocks
  int qrow[PQB];
  v8h qb[PQB][4];
#pragma unroll
  for (int j = 0; j < PQB; ++j) {
    qrow[j] = q0 + 16 * j + l15;
    const int qr_c = qrow[j] < len_q ? qrow[j] : (len_q - 1);
    const half_t* qp = p.q + (size_t)(q_begin + qr_c) * p.q_stride + (size_t)h * PDH + 8 * l4;
#pragma unroll
    for (int s = 0; s < 4; ++s) qb[j][s] = *reinterpret_cast<const v8h*>(qp + 32 * s);
  }
  v4f oacc[PQB][8];
  float m_run[PQB], l_run[PQB];
#pragma unroll
  for (int j = 0; j < PQB; ++j) {
    m_run[j] = -1e30f; l_run[j] = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) oacc[j][c] = (v4f){0.f, 0.f, 0.f, 0.f};
  }

  // key range this workgroup needs
  const int k_hi = p.causal ? min(len_k, q_last + off + 1) : len_k;             // exclusive
  const int win_lo = streaming ? first_local_key(q_first + off, local, gl) : 0;   // first local key of the first row
  auto skipped = [&](int kb) { return streaming && kb >= sink && kb + PKT <= win_lo; };   // tile inside the masked band

  // Tile staging by LDS-DMA (global_load_lds_dwordx4: one 1-KiB piece = 4 key rows per wave-instruction, no staging
  // registers, no ds_write pass).  The LDS image of an instruction is lane-linear (base + 16*lane), so the bank
  // swizzles are applied to the SOURCE address: lane l of piece i fills (row 4i + l/16, physical 16-B slot l%16)
  // and fetches the logical slot  l%16 ^ (row & 15)  of a K row  (conflict-free ds_read_b128 of the A operand)
  //                           or  l%16 ^ 2*(row & 7) of a V row  (conflict-free ds_read_b64_tr_b16: the 8 rows a
  // 32-lane group touches then sit in 8 disjoint 8-bank groups although the row pitch is 256 B).
  // Addresses are rebuilt per piece from a row index (v_add, v_min), ONE v_mad_u64_u32 (row x 32-bit token stride in
  // bytes + the uniform base) and the swizzle (2 ops): keeping eight 64-bit per-lane pointers alive across the tile
  // loop instead gets them spilled, and the reload's vmcnt(0) then serialises the DMA issue.
  const uint8_t* kbase = reinterpret_cast<const uint8_t*>(p.k + (size_t)k_begin * p.k_stride + (size_t)hk * PDH);
  const uint8_t* vbase = reinterpret_cast<const uint8_t*>(p.v + (size_t)k_begin * p.v_stride + (size_t)hk * PDH);
  const uint32_t kstride_b = (uint32_t)(p.k_stride * 2), vstride_b = (uint32_t)(p.v_stride * 2);   // host: < 2^32
#define PREFILL_DMA16(src_, dst_) lds_dma16((src_), (dst_))
#define PREFILL_DMA_TILE(kb_, kt_, vt_)                                                                           \
  do {                                                                                                            \
    int ln_ = lane;                                                                                               \
    asm volatile("" : "+v"(ln_));   /* opaque per tile: the per-piece lane constants must not be hoisted */         \
    const int ss_ = ln_ & 15, lr_ = ln_ >> 4;                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < PPT; ++i_) {                                                          \
      const int row_ = 4 * PPT * wave + 4 * i_ + lr_;                                                             \
      const uint32_t kr_ = (uint32_t)((kb_) + row_ < len_k ? (kb_) + row_ : (len_k - 1));                         \
      PREFILL_DMA16(kbase + ((uint64_t)kr_ * kstride_b + (uint32_t)((ss_ ^ (row_ & 15)) << 4)),                   \
                    (kt_) + (4 * PPT * wave + 4 * i_) * PKROW);                                                   \
      PREFILL_DMA16(vbase + ((uint64_t)kr_ * vstride_b + (uint32_t)((ss_ ^ (2 * (row_ & 7))) << 4)),              \
                    (vt_) + (4 * PPT * wave + 4 * i_) * PVROW);                                                   \
    }                                                                                                             \
  } while (0)
  auto next_tile = [&](int kb) {     // first tile >= kb that is not skipped (or >= k_hi)
    while (kb < k_hi && skipped(kb)) kb += PKT;
    return kb;
  };

  // V^T operand: row 4*l4 + l15/4 (+16, +32, +48 by immediates), 8 B at logical offset 32c + 8*(l15&3); the physical
  // 32-B block of logical block c is c ^ (row & 7)
  const int trow = 4 * l4 + (l15 >> 2);
  const int tr_a0 = (trow * PVROW + (l15 & 3) * 8) | ((trow & 7) << 5);   // address of logical block c: tr_a0 ^ (c << 5)
  int kb = next_tile(0);
  if (kb < k_hi) PREFILL_DMA_TILE(kb, ktile0, vtile0);
  __syncthreads();     // (carries the vmcnt(0) of the DMA pieces)
  auto tile_step = [&](auto parity) {
    constexpr int B = decltype(parity)::value;
    const int kb_next = next_tile(kb + PKT);
    const uint8_t* kt = B ? ktile1 : ktile0;
    const uint8_t* vt = B ? vtile1 : vtile0;
    // ---- S^T tile: 4 blocks of 16 keys x 2 query blocks ---------------------------------------------------
    // (MFMA phases run at raised priority: a wave that has matrix work ready goes ahead of the waves in their softmax)
    __builtin_amdgcn_s_setprio(1);
    v4f st[PQB][4];
#pragma unroll
    for (int j = 0; j < PQB; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) st[j][u] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = 16 * u + l15;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const v8h a = *reinterpret_cast<const v8h*>(kt + key * PKROW + (((4 * s + l4) ^ l15) << 4));
#pragma unroll
        for (int j = 0; j < PQB; ++j) st[j][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qb[j][s], st[j][u], 0, 0, 0);
      }
    }
    // st[j][u][r] = K[kb + 16u + 4*l4 + r] . Q[qrow[j]]
    // The next tile starts its trip global -> LDS (other buffer: every wave left it at the previous barrier) only
    // now: while a DMA is in flight hipcc waits lgkmcnt(0) in front of every consumer of a ds_read (it books the DMA
    // as a second kind of LDS event), which would serialise the 16 operand reads above against their MFMAs; the
    // softmax and the P.V phase below (whose reads are batched by hand) cover the flight time.
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
#if OMNI_PREFILL_ABLATE_DMA == 2     // timing experiment (wrong results): every tile re-fetches the first tile (L2-hot)
    if (kb_next < k_hi) PREFILL_DMA_TILE(0, B ? ktile0 : ktile1, B ? vtile0 : vtile1);
#elif !OMNI_PREFILL_ABLATE_DMA
    if (kb_next < k_hi) PREFILL_DMA_TILE(kb_next, B ? ktile0 : ktile1, B ? vtile0 : vtile1);
#endif
    __builtin_amdgcn_sched_barrier(0);
    // does any (row, key) pair of this workgroup's tile need the mask?  (workgroup-uniform)
    bool full = (kb + PKT <= len_k) && (q_first + PQROWS <= len_q);
    if (p.causal) full = full && (kb + PKT - 1 <= q_first + off);
    if (streaming) full = full && ((kb + PKT <= sink) || (kb >= first_local_key(q_last + off, local, gl)));
    // ---- online softmax (per query row = per lane column), probabilities straight into the B operand ----------
    // exp2 domain: p = exp2(s*scale2 - m); the maximum is taken on the raw scores (scale2 > 0) and the scaling rides
    // in the FMA that subtracts it.  Two code paths (workgroup-uniform): tiles that need no mask carry no predicate.
    v8h pb[PQB][2];
#pragma unroll
    for (int j = 0; j < PQB; ++j) {
      float tmax = -1e30f, m_new, alpha, psum = 0.0f;
      if (full) {     // one self-contained path per case: no register copies where the two would merge
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) tmax = __builtin_fmaxf(tmax, st[j][u][r]);
        tmax = rows4_max(tmax);   // the 4 lanes of a query row
        m_new = __builtin_fmaxf(m_run[j], tmax * scale2);
        alpha = __builtin_amdgcn_exp2f(m_run[j] - m_new);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(st[j][e >> 2][e & 3], scale2, -m_new));
          pb[j][e >> 3][e & 7] = (half_t)pe;   // keys 4*l4+r of blocks (2kk, 2kk+1) -> k-slots of the 32-key step kk
          psum += pe;
        }
      } else {
        bool okv[16];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kb + 16 * u + 4 * l4 + r;     // branch-free predicate (short-circuit && became branches)
            const int qpos = qrow[j] + off;
            int ok = (int)(key < len_k) & (int)(qrow[j] < len_q);
            ok &= (int)(!p.causal) | (int)(key <= qpos);
            ok &= (int)(!streaming) | (int)(key < sink) | (int)(key >= first_local_key(qpos, local, gl));
            okv[4 * u + r] = ok != 0;
            tmax = __builtin_fmaxf(tmax, ok ? st[j][u][r] : -1e30f);
          }
        tmax = rows4_max(tmax);
        m_new = __builtin_fmaxf(m_run[j], tmax * scale2);   // tmax = -1e30 (all masked) stays hugely negative
        alpha = __builtin_amdgcn_exp2f(m_run[j] - m_new);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float pe = okv[e] ? __builtin_amdgcn_exp2f(__builtin_fmaf(st[j][e >> 2][e & 3], scale2, -m_new)) : 0.0f;
          pb[j][e >> 3][e & 7] = (half_t)pe;
          psum += pe;
        }
      }
      l_run[j] = l_run[j] * alpha + psum;
      if (__builtin_amdgcn_ballot_w64(m_new != m_run[j]) != 0) {   // (wave-uniform) rescale only when a row's max moved
#pragma unroll
        for (int c = 0; c < 8; ++c) oacc[j][c] *= alpha;
      }
      m_run[j] = m_new;
    }
    // ---- O^T += V^T P^T ------------------------------------------------------------------------------------------
    // 16 steps (kk, c), each = two transposed reads (the A operand V^T) + one MFMA.  The schedule is pinned with
    // sched_group_barrier: the reads run three steps ahead of the MFMAs (three operand quads in flight, counted
    // lgkmcnt waits).  Left alone the compiler reuses ONE operand quad: read -> lgkmcnt(0) -> MFMA sixteen times per
    // tile (PMC: 44 % of the wave cycles parked on counters).
    __builtin_amdgcn_s_setprio(1);
    {
      v8h a16[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kk = i >> 3, c = i & 7;
        // the xor is recomputed per step (volatile asm): hoisted out of the tile loop the eight addresses cost eight
        // VGPRs, which pushed the kernel over the 128 that four waves per SIMD allow
        const uint8_t* src = vt + (32 * kk) * PVROW + (tr_a0 ^ (c << 5));
        const pv4hp lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (__attribute__((address_space(3))) pv4hp*)(__attribute__((address_space(3))) void*)(src));
        const pv4hp hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (__attribute__((address_space(3))) pv4hp*)(__attribute__((address_space(3))) void*)(src + 16 * PVROW));
        a16[i] = (v8h){(half_t)lo[0], (half_t)lo[1], (half_t)lo[2], (half_t)lo[3],
                       (half_t)hi[0], (half_t)hi[1], (half_t)hi[2], (half_t)hi[3]};
      }
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < PQB; ++j)
          oacc[j][i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[i], pb[j][i >> 3], oacc[j][i & 7], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);          // DS reads of steps 0..2
#pragma unroll
      for (int i = 0; i < 13; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, PQB, 0);      // MFMA of step i
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);        // DS reads of step i + 3
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 3 * PQB, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    // ---- the next tile has landed (vmcnt(0) of this wave's DMA pieces rides in the barrier) ----------------------
    __syncthreads();
    kb = kb_next;
  };
  while (kb < k_hi) {
    tile_step(IntTag<0>{});
    if (kb >= k_hi) break;
    tile_step(IntTag<1>{});
  }
  // ---- finish: row sum over the 4 lanes of a query row, normalise, store ------------------------------------
#pragma unroll
  for (int j = 0; j < PQB; ++j) {
    float l = l_run[j];
    l = rows4_sum(l);
    if (qrow[j] >= len_q) continue;
    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
    half_t* op = p.out + ((size_t)(q_begin + qrow[j]) * p.num_heads + h) * PDH + 4 * l4;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      typedef _Float16 v4h_t __attribute__((ext_vector_type(4)));
      const v4h_t o = {(half_t)(oacc[j][c][0] * inv), (half_t)(oacc[j][c][1] * inv), (half_t)(oacc[j][c][2] * inv),
                       (half_t)(oacc[j][c][3] * inv)};
      *reinterpret_cast<v4h_t*>(op + c * 16) = o;
    }
  }
}

#undef PREFILL_DMA_TILE

// ------------------------------------------------------------------------------------------------------------------------
// Ping-pong form of the 16-row kernel (omni_prefill_set_variant(2)).  Same tiles, layouts and arithmetic; what changes is WHEN the
// two halves of the workgroup do what.  In the form above all eight waves run S -> softmax -> P.V in step behind one barrier per
// tile, so on every SIMD the softmax's VALU issue and the MFMAs of the wave pair ADD UP instead of overlapping (profiles/r02_f:
// MFMA busy 40 % + VALU issue 57 %).  Here a tile is two segments behind two barriers, and waves 4-7 run one segment behind
// waves 0-3 (wave w and wave w + 4 share SIMD w):
//       segment 2i + 1                        segment 2i + 2
//   waves 0-3:  P.V(i), S(i+1)        |  softmax(i+1)                      (matrix | VALU)
//   waves 4-7:  softmax(i)            |  P.V(i), S(i+1)                    (VALU | matrix)
// so a SIMD always has one wave in a matrix segment beside one in the softmax.  K(i+1) and V(i) are read in segments 2i + 1 and
// 2i + 2: with the two-slot rings the DMAs of V(i+1) and K(i+2) go out at the start of segment 2i + 1 (their slots -- V(i-1)'s and
// K(i)'s -- were last read in segment 2i) and are waited for (vmcnt(0), every wave its own pieces) in front of the barrier that
// opens segment 2i + 3: one tile of flight, as above.  The DMAs are hidden from hipcc (lds_dma16_untracked): it otherwise waits lgkmcnt(0) in front of every consumer of a
// ds_read while one is in flight, and here the operand reads of P.V and S run beside the flight.
#define PP_DMA_ROWS(base_, stride_, swz_, kb_, t_, pitch_)                                                          \
  do {                                                                                                              \
    int ln_ = lane;                                                                                                 \
    asm volatile("" : "+v"(ln_));                                                                                   \
    const int ss_ = ln_ & 15, lr_ = ln_ >> 4;                                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < PPT; ++i_) {                                                            \
      const int row_ = 4 * PPT * wave + 4 * i_ + lr_;                                                               \
      const uint32_t kr_ = (uint32_t)((kb_) + row_ < len_k ? (kb_) + row_ : (len_k - 1));                           \
      lds_dma16_untracked((base_) + ((uint64_t)kr_ * (stride_) + (uint32_t)((ss_ ^ (swz_)) << 4)),                  \
                          (t_) + (4 * PPT * wave + 4 * i_) * (pitch_));                                             \
    }                                                                                                               \
  } while (0)
#define PP_DMA_K(kb_, kt_) PP_DMA_ROWS(kbase, kstride_b, (row_ & 15), kb_, kt_, PKROW)
#define PP_DMA_V(kb_, vt_) PP_DMA_ROWS(vbase, vstride_b, (2 * (row_ & 7)), kb_, vt_, PVROW)

__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(4, 4), amdgpu_num_vgpr(128)))
void prefill_attn_pp_kernel(PrefillArgs p) {
  static_assert(PQB == 1 && PWAVES == 8, "the ping-pong form is written for 8 waves x 16 rows");
  __shared__ __attribute__((aligned(16))) uint8_t ktile0[PKTILE];
  __shared__ __attribute__((aligned(16))) uint8_t ktile1[PKTILE];
  __shared__ __attribute__((aligned(16))) uint8_t vtile0[PVTILE];
  __shared__ __attribute__((aligned(16))) uint8_t vtile1[PVTILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  int b = blockIdx.z, h = blockIdx.y, qt = gridDim.x - 1 - blockIdx.x;
  if (p.q_tiles > 0 && !prefill_map_block(p, b, h, qt)) return;
  const int hk = h / (p.num_heads / p.num_kv_heads);
  const int q_begin = p.cu_q[b], len_q = p.cu_q[b + 1] - q_begin;
  const int k_begin = p.cu_k[b], len_k = p.cu_k[b + 1] - k_begin;
  const int q_first = qt * PQROWS;
  if (q_first >= len_q) return;
  const int q_last = min(q_first + PQROWS, len_q) - 1;
  const int q0 = q_first + wave * 16;
  const int off = len_k - len_q;
  const bool streaming = p.head_mask_type != nullptr && p.head_mask_type[h] < 0;
  const int gl = p.gran_log2;
  const int sink = streaming ? p.streaming_info[2 * h] << gl : 0;
  const int local = streaming ? p.streaming_info[2 * h + 1] : 0;
  const float scale2 = 0.08838834764831845f * 1.4426950408889634f;
  const int qrow = q0 + l15;
  v8h qb[4];
  {
    const int qr_c = qrow < len_q ? qrow : (len_q - 1);
    const half_t* qp = p.q + (size_t)(q_begin + qr_c) * p.q_stride + (size_t)h * PDH + 8 * l4;
#pragma unroll
    for (int s = 0; s < 4; ++s) qb[s] = *reinterpret_cast<const v8h*>(qp + 32 * s);
  }
  v4f oacc[8];
  float m_run = -1e30f, l_run = 0.0f;
#pragma unroll
  for (int c = 0; c < 8; ++c) oacc[c] = (v4f){0.f, 0.f, 0.f, 0.f};
  const int k_hi = p.causal ? min(len_k, q_last + off + 1) : len_k;
  const int win_lo = streaming ? first_local_key(q_first + off, local, gl) : 0;
  auto skipped = [&](int kb) { return streaming && kb >= sink && kb + PKT <= win_lo; };
  auto next_tile = [&](int kb) {
    while (kb < k_hi && skipped(kb)) kb += PKT;
    return kb;
  };
  const uint8_t* kbase = reinterpret_cast<const uint8_t*>(p.k + (size_t)k_begin * p.k_stride + (size_t)hk * PDH);
  const uint8_t* vbase = reinterpret_cast<const uint8_t*>(p.v + (size_t)k_begin * p.v_stride + (size_t)hk * PDH);
  const uint32_t kstride_b = (uint32_t)(p.k_stride * 2), vstride_b = (uint32_t)(p.v_stride * 2);
  const int trow = 4 * l4 + (l15 >> 2);
  const int tr_a0 = (trow * PVROW + (l15 & 3) * 8) | ((trow & 7) << 5);

  // ---- the three stages of a tile (the 16-row kernel's own code) ---------------------------------------------------------
  auto stage_s = [&](const uint8_t* kt, v4f (&st)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) st[u] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = 16 * u + l15;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const v8h a = *reinterpret_cast<const v8h*>(kt + key * PKROW + (((4 * s + l4) ^ l15) << 4));
        st[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qb[s], st[u], 0, 0, 0);
      }
    }
  };
  auto stage_softmax = [&](int kb, const v4f (&st)[4], v8h (&pb)[2]) {
    bool full = (kb + PKT <= len_k) && (q_first + PQROWS <= len_q);
    if (p.causal) full = full && (kb + PKT - 1 <= q_first + off);
    if (streaming) full = full && ((kb + PKT <= sink) || (kb >= first_local_key(q_last + off, local, gl)));
    float tmax = -1e30f, m_new, alpha, psum = 0.0f;
    if (full) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) tmax = __builtin_fmaxf(tmax, st[u][r]);
      tmax = rows4_max(tmax);
      m_new = __builtin_fmaxf(m_run, tmax * scale2);
      alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(st[e >> 2][e & 3], scale2, -m_new));
        pb[e >> 3][e & 7] = (half_t)pe;
        psum += pe;
      }
    } else {
      bool okv[16];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kb + 16 * u + 4 * l4 + r;
          const int qpos = qrow + off;
          int ok = (int)(key < len_k) & (int)(qrow < len_q);
          ok &= (int)(!p.causal) | (int)(key <= qpos);
          ok &= (int)(!streaming) | (int)(key < sink) | (int)(key >= first_local_key(qpos, local, gl));
          okv[4 * u + r] = ok != 0;
          tmax = __builtin_fmaxf(tmax, ok ? st[u][r] : -1e30f);
        }
      tmax = rows4_max(tmax);
      m_new = __builtin_fmaxf(m_run, tmax * scale2);
      alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pe = okv[e] ? __builtin_amdgcn_exp2f(__builtin_fmaf(st[e >> 2][e & 3], scale2, -m_new)) : 0.0f;
        pb[e >> 3][e & 7] = (half_t)pe;
        psum += pe;
      }
    }
    l_run = l_run * alpha + psum;
    if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c) oacc[c] *= alpha;
    }
    m_run = m_new;
  };
  auto stage_pv = [&](const uint8_t* vt, const v8h (&pb)[2]) {
    v8h a16[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kk = i >> 3, c = i & 7;
      const uint8_t* src = vt + (32 * kk) * PVROW + (tr_a0 ^ (c << 5));
      const pv4hp lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
          (__attribute__((address_space(3))) pv4hp*)(__attribute__((address_space(3))) void*)(src));
      const pv4hp hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
          (__attribute__((address_space(3))) pv4hp*)(__attribute__((address_space(3))) void*)(src + 16 * PVROW));
      a16[i] = (v8h){(half_t)lo[0], (half_t)lo[1], (half_t)lo[2], (half_t)lo[3],
                     (half_t)hi[0], (half_t)hi[1], (half_t)hi[2], (half_t)hi[3]};
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[i], pb[i >> 3], oacc[i & 7], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int i = 0; i < 13; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
  };
  // barrier that opens a B segment: this wave's DMA pieces have landed, then everybody's
  auto barrier_landed = [&]() {
    lds_dma_wait_all();
    __builtin_amdgcn_s_barrier();
  };

  // ---- prologue: K(0), V(0), K(1) ---------------------------------------------------------------------------------------
  int kc = next_tile(0);                                     // tile i
  int kn = kc < k_hi ? next_tile(kc + PKT) : k_hi;           // tile i + 1
  int kn2 = kn < k_hi ? next_tile(kn + PKT) : k_hi;          // tile i + 2
  if (kc < k_hi) { PP_DMA_K(kc, ktile0); PP_DMA_V(kc, vtile0); }
  if (kn < k_hi) PP_DMA_K(kn, ktile1);
  barrier_landed();
  v4f st[4];
  v8h pb[2];
  const bool first_half = wave < PWAVES / 2;
  // ONE instruction stream for both halves -- S(0); then per tile: softmax(i) | barrier | P.V(i), S(i+1) | barrier -- with waves
  // 4-7 held back by one extra barrier behind S(0) (waves 0-3 run the matching one behind their last tile): from then on every
  // barrier pairs a softmax segment of one half with a matrix segment of the other.  What differs per half is only where the
  // DMAs are issued and waited for (both at the same points of the WORKGROUP's timeline, see the kernel's header).
  auto issue_dma = [&](uint8_t* v_slot, uint8_t* k_slot) {       // V(i + 1) -> the slot of V(i - 1), K(i + 2) -> the slot of K(i)
    if (kn < k_hi) PP_DMA_V(kn, v_slot);
    if (kn2 < k_hi) PP_DMA_K(kn2, k_slot);
  };
  auto tile = [&](auto parity) {
    constexpr int B = decltype(parity)::value;
    uint8_t* kt_cur = B ? ktile1 : ktile0;
    uint8_t* kt_nxt = B ? ktile0 : ktile1;
    uint8_t* vt_cur = B ? vtile1 : vtile0;
    uint8_t* vt_oth = B ? vtile0 : vtile1;
    if (!first_half) issue_dma(vt_oth, kt_cur);
    __builtin_amdgcn_sched_barrier(0);
    stage_softmax(kc, st, pb);
    __builtin_amdgcn_sched_barrier(0);
    if (first_half) lds_dma_wait_all();
    __builtin_amdgcn_s_barrier();
    if (first_half) issue_dma(vt_oth, kt_cur);
    __builtin_amdgcn_sched_barrier(0);
    stage_pv(vt_cur, pb);                                    // P.V(i)
    __builtin_amdgcn_sched_barrier(0);
    if (kn < k_hi) stage_s(kt_nxt, st);                      // S(i + 1)
    __builtin_amdgcn_sched_barrier(0);
    if (!first_half) lds_dma_wait_all();
    __builtin_amdgcn_s_barrier();
    kc = kn; kn = kn2;
    kn2 = kn2 < k_hi ? next_tile(kn2 + PKT) : k_hi;
  };
  if (kc < k_hi) {
    stage_s(ktile0, st);                                     // S(0)
    __builtin_amdgcn_sched_barrier(0);
    if (!first_half) __builtin_amdgcn_s_barrier();
    while (true) {
      tile(IntTag<0>{});
      if (kc >= k_hi) break;
      tile(IntTag<1>{});
      if (kc >= k_hi) break;
    }
    if (first_half) __builtin_amdgcn_s_barrier();
  }
  // ---- finish ----
  {
    const float l = rows4_sum(l_run);
    if (qrow < len_q) {
      const float inv = l > 0.0f ? 1.0f / l : 0.0f;
      half_t* op = p.out + ((size_t)(q_begin + qrow) * p.num_heads + h) * PDH + 4 * l4;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        typedef _Float16 v4h_t __attribute__((ext_vector_type(4)));
        const v4h_t o = {(half_t)(oacc[c][0] * inv), (half_t)(oacc[c][1] * inv), (half_t)(oacc[c][2] * inv),
                         (half_t)(oacc[c][3] * inv)};
        *reinterpret_cast<v4h_t*>(op + c * 16) = o;
      }
    }
  }
}
#undef PP_DMA_K
#undef PP_DMA_V
#undef PP_DMA_ROWS

// ------------------------------------------------------------------------------------------------------------------------
// 32-row form (v_mfma_f32_32x32x16_f16).  Same tiles, same LDS-DMA staging, same online softmax; what changes is the
// shape of a wave's work: 32 query rows x 64 keys per tile instead of 16 x 64.
//   * S^T = K Q^T per 32-key block: A = K rows from LDS (lane l: key l%32, 16 B of dims 16s + 8*(l/32)), B = Q in
//     registers (lane l: query l%32, the same dims).  D: lane l holds ONE query (column l%32) and the 16 keys
//     (r&3) + 8*(r>>2) + 4*(l/32) of the block: the row maximum is 31 in-lane max + one permlane32 swap.
//   * P^T is the B operand of O^T = V^T P^T as it stands: a k-step of the second product may enumerate its 16 keys in any
//     order as long as A agrees, so k-step j of a block takes the lane's own values r = 8j .. 8j+7 (keys 16j + 4*(l/32) +
//     {0..3} and + 8) -- no cross-lane exchange, just cvt_pk.  A = V^T: two ds_read_b64_tr_b16 per operand (rows
//     16j + 4*(l/32) + {0..3}, and + 8), each 16-lane group transposing a [4 keys][16 dims] block.
//   * per wave and tile: 32 MFMAs of 32 cycles for 2048 scores (the 16-row form: 2 x 32 MFMAs of 16 cycles), the K and V
//     operands are read from LDS once per 32 rows instead of once per 16, and the bookkeeping per row (maximum, rescale,
//     addresses, DMA issue) is amortised over twice the flops.
// V tile swizzle of this kernel: physical 64-B block = logical block ^ (row & 3) (the four rows a 32-lane half of a
// transposed read touches sit in four different bank quarters); K tile swizzle as above (slot ^ (row & 15)).
#ifndef OMNI_PREFILL32_WAVES
#define OMNI_PREFILL32_WAVES 4
#endif
constexpr int P32W = OMNI_PREFILL32_WAVES;
constexpr int P32ROWS = 32 * P32W;                       // query rows per workgroup
constexpr int P32PPT = (PKT * 16) / (64 * P32W);         // 1-KiB DMA pieces of a K (or V) tile per wave
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(64 * P32W, 8 / P32W) __attribute__((amdgpu_waves_per_eu(2, 2)))
void prefill_attn32_kernel(PrefillArgs p) {
  __shared__ __attribute__((aligned(16))) uint8_t ktile0[PKTILE];
  __shared__ __attribute__((aligned(16))) uint8_t ktile1[PKTILE];
  __shared__ __attribute__((aligned(16))) uint8_t vtile0[PVTILE];
  __shared__ __attribute__((aligned(16))) uint8_t vtile1[PVTILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, hi = lane >> 5, l15 = lane & 15, grp = lane >> 4;
  int b = blockIdx.z, h = blockIdx.y, qt = gridDim.x - 1 - blockIdx.x;
  if (p.q_tiles > 0 && !prefill_map_block(p, b, h, qt)) return;
  const int hk = h / (p.num_heads / p.num_kv_heads);
  const int q_begin = p.cu_q[b], len_q = p.cu_q[b + 1] - q_begin;
  const int k_begin = p.cu_k[b], len_k = p.cu_k[b + 1] - k_begin;
  const int q_first = qt * P32ROWS;
  if (q_first >= len_q) return;
  const int q_last = min(q_first + P32ROWS, len_q) - 1;
  const int off = len_k - len_q;
  const bool streaming = p.head_mask_type != nullptr && p.head_mask_type[h] < 0;
  const int gl = p.gran_log2;
  const int sink = streaming ? p.streaming_info[2 * h] << gl : 0;      // in tokens
  const int local = streaming ? p.streaming_info[2 * h + 1] : 0;       // in units of 1 << gl tokens
  const float scale2 = 0.08838834764831845f * 1.4426950408889634f;
  const int qrow = q_first + wave * 32 + l32;               // this lane's query row (lanes l and l + 32 share it)

  // B operand of S^T: Q[qrow][16s + 8*hi + (0..7)], s = 0..7
  v8h qb[8];
  {
    const int qr_c = qrow < len_q ? qrow : (len_q - 1);
    const half_t* qp = p.q + (size_t)(q_begin + qr_c) * p.q_stride + (size_t)h * PDH + 8 * hi;
#pragma unroll
    for (int s = 0; s < 8; ++s) qb[s] = *reinterpret_cast<const v8h*>(qp + 16 * s);
  }
  v16f oacc[4];      // O^T: dims 32*c + (r&3) + 8*(r>>2) + 4*hi of query l32
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[c][r] = 0.0f;
  float m_run = -1e30f, l_run = 0.0f;

  const int k_hi = p.causal ? min(len_k, q_last + off + 1) : len_k;
  const int win_lo = streaming ? first_local_key(q_first + off, local, gl) : 0;
  auto skipped = [&](int kb) { return streaming && kb >= sink && kb + PKT <= win_lo; };
  auto next_tile = [&](int kb) {
    while (kb < k_hi && skipped(kb)) kb += PKT;
    return kb;
  };
  const uint8_t* kbase = reinterpret_cast<const uint8_t*>(p.k + (size_t)k_begin * p.k_stride + (size_t)hk * PDH);
  const uint8_t* vbase = reinterpret_cast<const uint8_t*>(p.v + (size_t)k_begin * p.v_stride + (size_t)hk * PDH);
  const uint32_t kstride_b = (uint32_t)(p.k_stride * 2), vstride_b = (uint32_t)(p.v_stride * 2);
#define PREFILL32_DMA_TILE(kb_, kt_, vt_)                                                                         \
  do {                                                                                                            \
    int ln_ = lane;                                                                                               \
    asm volatile("" : "+v"(ln_));                                                                                 \
    const int ss_ = ln_ & 15, lr_ = ln_ >> 4;                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < P32PPT; ++i_) {                                                       \
      const int row_ = 4 * P32PPT * wave + 4 * i_ + lr_;                                                          \
      const uint32_t kr_ = (uint32_t)((kb_) + row_ < len_k ? (kb_) + row_ : (len_k - 1));                         \
      lds_dma16_untracked(kbase + ((uint64_t)kr_ * kstride_b + (uint32_t)((ss_ ^ (row_ & 15)) << 4)),                       \
                (kt_) + (4 * P32PPT * wave + 4 * i_) * PKROW);                                                    \
      lds_dma16_untracked(vbase + ((uint64_t)kr_ * vstride_b + (uint32_t)((ss_ ^ ((row_ & 3) << 2)) << 4)),                 \
                (vt_) + (4 * P32PPT * wave + 4 * i_) * PVROW);                                                    \
    }                                                                                                             \
  } while (0)

  // LDS addresses.  K operand (key 32u + l32, logical 16-B slot 2s + hi): physical slot (2s + hi) ^ l15.
  int kaddr[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) kaddr[s] = l32 * PKROW + (((2 * s + hi) ^ l15) << 4);
  // V^T operand: row 4*hi + (l15 >> 2) (+ 16j + 32u, + 8 by immediates), dims 32c + 16*(grp & 1) + 4*(l15 & 3);
  // (row & 3) = l15 >> 2, so the physical 64-B block of logical block c is c ^ (l15 >> 2)
  int vaddr[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
    vaddr[c] = (4 * hi + (l15 >> 2)) * PVROW + ((c ^ (l15 >> 2)) << 6) + 32 * (grp & 1) + 8 * (l15 & 3);

  // (the tile DMAs of this kernel are hidden from hipcc -- inline asm -- and waited for by hand: while it knows of a DMA in
  //  flight it puts a full lgkmcnt(0) in front of every consumer of a ds_read, which breaks the read pipelines below)
  int kb = next_tile(0);
  if (kb < k_hi) PREFILL32_DMA_TILE(kb, ktile0, vtile0);
  lds_dma_wait_all();
  __syncthreads();
  auto tile_step = [&](auto parity, auto fulltag) {
    constexpr int B = decltype(parity)::value;
    constexpr bool FULL = decltype(fulltag)::value;
    const int kb_next = next_tile(kb + PKT);
    const uint8_t* kt = B ? ktile1 : ktile0;
    const uint8_t* vt = B ? vtile1 : vtile0;
    // ---- S^T: two 32-key blocks x 8 dim steps ---------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(1);
    v16f st[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[u][r] = 0.0f;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const v8h a = *reinterpret_cast<const v8h*>(kt + 32 * u * PKROW + kaddr[s]);
        st[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qb[s], st[u], 0, 0, 0);
      }
    }
    // operand reads four MFMAs (128 cycles) ahead
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
    if (kb_next < k_hi) PREFILL32_DMA_TILE(kb_next, B ? ktile0 : ktile1, B ? vtile0 : vtile1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- online softmax: this lane's 32 scores of query l32 ------------------------------------------------------
    float tmax = -1e30f;
    bool okv[2][16];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if constexpr (FULL) {
          okv[u][r] = true;
          tmax = __builtin_fmaxf(tmax, st[u][r]);
        } else {
          const int key = kb + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const int qpos = qrow + off;
          int ok = (int)(key < len_k) & (int)(qrow < len_q);
          ok &= (int)(!p.causal) | (int)(key <= qpos);
          ok &= (int)(!streaming) | (int)(key < sink) | (int)(key >= first_local_key(qpos, local, gl));
          okv[u][r] = ok != 0;
          tmax = __builtin_fmaxf(tmax, ok ? st[u][r] : -1e30f);
        }
      }
    {      // the other 32 keys of the row live in lane l ^ 32
      unsigned x = __builtin_bit_cast(unsigned, tmax);
      auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
      tmax = __builtin_fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
    }
    const float m_new = __builtin_fmaxf(m_run, tmax * scale2);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.0f;
    v8h pb[2][2];      // [block][k-step]: the lane's own values r = 8j .. 8j + 7
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(st[u][r], scale2, -m_new));
        if constexpr (!FULL) pe = okv[u][r] ? pe : 0.0f;
        pb[u][r >> 3][r & 7] = (half_t)pe;
        psum += pe;
      }
    l_run = l_run * alpha + psum;
    if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[c][r] *= alpha;
    }
    m_run = m_new;
    // ---- O^T += V^T P^T: 4 k-steps (block u, step j) x 4 dim blocks ---------------------------------------------------
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int uj = 0; uj < 4; ++uj) {
      const int u = uj >> 1, j = uj & 1;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint8_t* src = vt + (32 * u + 16 * j) * PVROW + vaddr[c];
        const pv4hp lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (__attribute__((address_space(3))) pv4hp*)(__attribute__((address_space(3))) void*)(src));
        const pv4hp hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (__attribute__((address_space(3))) pv4hp*)(__attribute__((address_space(3))) void*)(src + 8 * PVROW));
        const v8h a = {(half_t)lo[0], (half_t)lo[1], (half_t)lo[2], (half_t)lo[3],
                       (half_t)hi4[0], (half_t)hi4[1], (half_t)hi4[2], (half_t)hi4[3]};
        oacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb[u][j], oacc[c], 0, 0, 0);
      }
    }
    // the transposed reads run three operands ahead of the MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int i = 0; i < 13; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    __builtin_amdgcn_s_setprio(0);
    lds_dma_wait_all();       // this wave's pieces of the next tile have landed; the barrier covers everybody's
    __syncthreads();
    kb = kb_next;
  };
  auto is_full = [&](int kb_) {
    bool full = (kb_ + PKT <= len_k) && (q_first + P32ROWS <= len_q);
    if (p.causal) full = full && (kb_ + PKT - 1 <= q_first + off);
    if (streaming) full = full && ((kb_ + PKT <= sink) || (kb_ >= first_local_key(q_last + off, local, gl)));
    return full;
  };
  while (kb < k_hi) {
    if (is_full(kb)) tile_step(IntTag<0>{}, IntTag<1>{}); else tile_step(IntTag<0>{}, IntTag<0>{});
    if (kb >= k_hi) break;
    if (is_full(kb)) tile_step(IntTag<1>{}, IntTag<1>{}); else tile_step(IntTag<1>{}, IntTag<0>{});
  }
  // ---- finish: the row sum lives in two lanes; normalise, store (4 consecutive dims per register quad) ---------------
  {
    unsigned x = __builtin_bit_cast(unsigned, l_run);
    auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    l_run = __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
  }
  if (qrow >= len_q) return;
  const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
  half_t* op = p.out + ((size_t)(q_begin + qrow) * p.num_heads + h) * PDH + 4 * hi;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      typedef _Float16 v4h_t __attribute__((ext_vector_type(4)));
      const v4h_t o = {(half_t)(oacc[c][4 * g + 0] * inv), (half_t)(oacc[c][4 * g + 1] * inv),
                       (half_t)(oacc[c][4 * g + 2] * inv), (half_t)(oacc[c][4 * g + 3] * inv)};
      *reinterpret_cast<v4h_t*>(op + 32 * c + 8 * g) = o;
    }
}
#undef PREFILL32_DMA_TILE

}  // namespace omni

using namespace omni;

static thread_local int g_prefill_variant = OMNI_PREFILL_MFMA32;
// Tuning / test hook: 0 = the 16-row form (default), 1 = the 32-row form.  Same results within the attention tolerance.
extern "C" void omni_prefill_set_variant(int variant) { g_prefill_variant = (variant == 1 || variant == 2) ? variant : 0; }
#ifndef OMNI_PREFILL_XCD_SPLIT_MASKED
#define OMNI_PREFILL_XCD_SPLIT_MASKED 8
#endif
static thread_local int g_prefill_xcd_split = 0;
// Tuning hook: XCDs that share one kv head's query tiles when streaming heads are present (0 = default, else 1 / 2 / 4 / 8).
extern "C" void omni_prefill_set_xcd_split(int w) { g_prefill_xcd_split = (w == 1 || w == 2 || w == 4 || w == 8) ? w : 0; }

static int prefill_attention_common(void* out_f16, const void* q_f16, const void* k_f16, const void* v_f16,
                                    int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                    const void* cu_seqlens_q_i32, const void* cu_seqlens_k_i32, int batch,
                                    int max_seqlen_q, int num_heads, int num_kv_heads, int head_dim, int causal,
                                    const void* head_mask_type_i32, const void* streaming_info_i32, int gran_log2, void* stream) {
  if (!out_f16 || !q_f16 || !k_f16 || !v_f16 || !cu_seqlens_q_i32 || !cu_seqlens_k_i32) return OMNI_EINVAL;
  if (head_dim != PDH || batch < 1 || num_heads < 1 || num_kv_heads < 1 || num_heads % num_kv_heads != 0 ||
      max_seqlen_q < 1 || q_stride % 8 != 0 || k_stride % 8 != 0 || v_stride % 8 != 0 || k_stride < 0 || v_stride < 0 ||
      k_stride >= (1LL << 31) || v_stride >= (1LL << 31))
    return OMNI_EINVAL;
  if ((head_mask_type_i32 == nullptr) != (streaming_info_i32 == nullptr)) return OMNI_EINVAL;
  PrefillArgs a;
  a.q = (const half_t*)q_f16; a.k = (const half_t*)k_f16; a.v = (const half_t*)v_f16; a.out = (half_t*)out_f16;
  a.q_stride = q_stride; a.k_stride = k_stride; a.v_stride = v_stride;
  a.cu_q = (const int*)cu_seqlens_q_i32; a.cu_k = (const int*)cu_seqlens_k_i32;
  a.head_mask_type = (const int*)head_mask_type_i32; a.streaming_info = (const int*)streaming_info_i32;
  a.num_heads = num_heads; a.num_kv_heads = num_kv_heads; a.causal = causal;
  a.gran_log2 = gran_log2;
  const bool form32 = g_prefill_variant == 1;
  const int rows_per_wg = form32 ? P32ROWS : PQROWS;
  const int q_tiles = (max_seqlen_q + rows_per_wg - 1) / rows_per_wg;
  dim3 grid(q_tiles, num_heads, batch);
  a.q_tiles = 0;
  a.xcd_split = 1;
  {
    // W: with streaming heads present the dense work has to be spread (see prefill_map_block): measured at 128 K tokens,
    // 4 dense + 4 streaming kv heads, W = 1 / 2 / 4 / 8: 143 / 92 / 92 / 87 ms (122 ms with whole kv heads pinned in index
    // order); all-dense launches keep whole kv heads on one XCD (145 vs 152 ms).  (Hk * W) must be a multiple of 8.
    int W = head_mask_type_i32 ? (g_prefill_xcd_split > 0 ? g_prefill_xcd_split : OMNI_PREFILL_XCD_SPLIT_MASKED) : 1;
    while ((num_kv_heads * W) % 8 != 0) W *= 2;
    const long long rows = (q_tiles + W - 1) / W;
    const long long wgs = 8LL * batch * rows * ((num_kv_heads * W) / 8) * (num_heads / num_kv_heads);
    if (W <= 8 && wgs < (1LL << 31)) {
      a.q_tiles = q_tiles;
      a.xcd_split = W;
      grid = dim3((unsigned)wgs, 1, 1);
    }
  }
  if (form32) hipLaunchKernelGGL(prefill_attn32_kernel, grid, dim3(64 * P32W), 0, (hipStream_t)stream, a);
  else if (g_prefill_variant == 2) hipLaunchKernelGGL(prefill_attn_pp_kernel, grid, dim3(64 * PWAVES), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(prefill_attn_kernel, grid, dim3(64 * PWAVES), 0, (hipStream_t)stream, a);
  return omni_launch_status();
}

extern "C" int omni_prefill_attention(void* out_f16, const void* q_f16, const void* k_f16, const void* v_f16,
                                      int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                      const void* cu_seqlens_q_i32, const void* cu_seqlens_k_i32, int batch,
                                      int max_seqlen_q, int num_heads, int num_kv_heads, int head_dim, int causal,
                                      const void* head_mask_type_i32, const void* streaming_info_i32, void* stream) {
  return prefill_attention_common(out_f16, q_f16, k_f16, v_f16, q_stride, k_stride, v_stride, cu_seqlens_q_i32, cu_seqlens_k_i32,
                                  batch, max_seqlen_q, num_heads, num_kv_heads, head_dim, causal, head_mask_type_i32,
                                  streaming_info_i32, 0, stream);
}

// block_sparse_attn.block_streaming_attn_func (ctx_attn_func.py:47-59): the same causal attention with streaming_info =
// (sink, local) per q head counted in BLOCKS of 128 tokens (the package's m / n block size): a streaming head's query in block i
// sees the first `sink` key blocks and the key blocks i - local + 1 .. i (causal inside its own).  Semantics inferred like the
// token form's (the package is not vendored; the reference's only call site is dead code).
extern "C" int omni_prefill_attention_block_streaming(void* out_f16, const void* q_f16, const void* k_f16, const void* v_f16,
                                                      int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                                      const void* cu_seqlens_q_i32, const void* cu_seqlens_k_i32, int batch,
                                                      int max_seqlen_q, int num_heads, int num_kv_heads, int head_dim,
                                                      const void* head_mask_type_i32, const void* streaming_info_i32,
                                                      void* stream) {
  if (!head_mask_type_i32 || !streaming_info_i32) return OMNI_EINVAL;
  return prefill_attention_common(out_f16, q_f16, k_f16, v_f16, q_stride, k_stride, v_stride, cu_seqlens_q_i32, cu_seqlens_k_i32,
                                  batch, max_seqlen_q, num_heads, num_kv_heads, head_dim, 1, head_mask_type_i32,
                                  streaming_info_i32, 7, stream);
}
