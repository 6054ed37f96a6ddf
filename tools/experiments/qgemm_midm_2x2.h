// Mid-M kernel (M = 33 .. 128 rows per tile): the decode GEMMs at the batches between the single-wave GEMV tiles (M <= 32)
// and the MFMA-bound prefill tile (M > 128).  Included by qgemm_kernel.h.
//
// At 64 .. 128 rows a weight byte meets 64 .. 128 activation bytes: the launch is bound by HBM AND close to the MFMA floor
// (Llama-3-8B gate_up at M = 128: 58.7 MB = 10.2 us of stream, 30 GOP = 7.6 us of int8 MFMA), so every byte may enter a
// CU once and every instruction has to count.  The single-wave tiles (32 / 64 rows, qgemm_kernel.h) re-read every weight
// byte from L1 / L2 per row tile and every activation byte per 64-channel group (4x the algorithmic bytes through a path
// that fills L1 at <= 38 B/clk: 36 us on that shape); the 128 x 256 prefill tile has too few tiles to pull on all 256 CUs
// without K slices whose int32 slabs cost more than the weights.  This kernel:
//
//   * workgroup = 128 output channels (two 64-channel groups) x all rows of the tile (MB x 16 <= 128) x one K slice
//     (the whole K where N / 128 tiles fill the chip: no slab), 8 waves = 2 groups x 4 K PHASES: wave (g, s) owns k-step
//     s of every 256-k chunk for group g and all rows.  A weight byte is fetched once, by one wave; the unpack runs once per
//     byte (48 VALU per 32 MFMAs at 128 rows -- the 32-row tiles spend 40 per 8);
//   * EVERYTHING the K loop reads arrives by LDS-DMA (global_load_lds, issued from inline asm, completion owned by explicit
//     vmcnt waits): the activation chunk (rows x 256 B, once per workgroup, NBUF buffers, one barrier per chunk; lane
//     transposition of the packed registers as in w4a8_gemm_exact_kernel, so that a DMA-written row is the B operand), the
//     wave's own 2 KiB of packed weights per chunk (a private NBUF-slot ring, read back with two ds_read_b128 as the MFMA A
//     operand source) and the per-group parameters.  Every request is quad-coalesced (see "request addressing").  Why not registers for the weights: the CU's
//     vector memory path returns in order, so an L2-hit activation piece queued behind an HBM-miss weight load lands with
//     HBM latency; every chunk needs its activation tile, so with the tile requested one chunk ahead (the first version:
//     weights in a register ring, counted compiler waits) a chunk took one loaded HBM latency -- 1.56 us against 0.49 us of
//     MFMA, 25 us per launch on the shape above (profiles/r05_a).  All requests of chunk c + D are now issued in chunk c
//     (D = NBUF - 1), and hipcc's own waitcnt pass sees no vector memory operation in the loop: it cannot fold a fresh DMA
//     piece into a counted wait for an old register load, and there is no loop-carried register that an async load targets;
//   * the four K-phase partials of a group meet in LDS after the K loop (two exchange rounds over the ring memory, static
//     accumulator indices only); wave (g, s) finishes row quarter s: epilogue to fp16, or the int32 slab of its K slice for
//     the slab consumers / splitk_epilogue_kernel.
// Activation traffic per launch = (N / 128) x M x K bytes from L2 (2x the weight bytes at M = 128), weights 1x from HBM.
// LDS-DMA reaches all 160 KiB of gfx950's LDS (tools/lds_dma_hi_probe.hip).
#pragma once

namespace omni {

#ifndef OMNI_MIDM_ABLATE
#define OMNI_MIDM_ABLATE 0        // timing experiments (WRONG results): 1 no requests behind the prologue, 2 one B-operand read per
#endif                            // chunk, 4 no unpack arithmetic, 8 no barrier in the K loop, 16 a quarter of the MFMAs
#ifndef OMNI_MIDM_PRIO
#define OMNI_MIDM_PRIO 0          // 1: alternate the favoured wave of each SIMD stage by stage (see the K loop; measured neutral)
#endif

// LDS-DMA statements of one wave.  A piece = one wave instruction: lane l's 16 B (dwordx4) or 4 B (dword) land at
// lds_dst + piece * (64 * bytes) + l * bytes.  Source = scalar base + the lane's 32-bit offset (+ an immediate).  M0
// (compiler-reserved) is saved once per statement and restored at its end.
#define OMNI_DMA_HEAD "s_nop 4\n\ts_mov_b32 %0, m0\n\t"
#define OMNI_DMA_TAIL "s_mov_b32 m0, %0"
// activation pieces: 4 rows x 256 B each
__device__ __forceinline__ void lds_dma16_x2(const void* sbase, uint32_t v0, uint32_t v1, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(OMNI_DMA_HEAD
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
      OMNI_DMA_TAIL
      : "=&s"(keep) : "s"(sbase), "v"(v0), "v"(v1), "s"(lds_dst) : "memory", "scc");
}
__device__ __forceinline__ void lds_dma16_x4(const void* sbase, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3,
                                             uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(OMNI_DMA_HEAD
      "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
      OMNI_DMA_TAIL
      : "=&s"(keep) : "s"(sbase), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(lds_dst) : "memory", "scc");
}
// packed int4 weights of one k-step: 1 KiB (two consecutive 512-B tiles) of each of the group's two tile rows.
// (No immediate offsets on LDS-DMA instructions: the instruction offset is added to the LDS address as well.)
template <bool NT>
__device__ __forceinline__ void lds_dma_w4(const void* sbase, uint32_t v0, uint32_t v1, uint32_t lds_dst) {
  uint32_t keep;
  if constexpr (NT)
    asm volatile(OMNI_DMA_HEAD
        "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1 nt\n\t"
        OMNI_DMA_TAIL
        : "=&s"(keep) : "s"(sbase), "v"(v0), "v"(v1), "s"(lds_dst) : "memory", "scc");
  else
    asm volatile(OMNI_DMA_HEAD
        "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
        OMNI_DMA_TAIL
        : "=&s"(keep) : "s"(sbase), "v"(v0), "v"(v1), "s"(lds_dst) : "memory", "scc");
}
// int8 weights of one k-step: four 16-row blocks
template <bool NT>
__device__ __forceinline__ void lds_dma_w8(const void* sbase, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t lds_dst) {
  uint32_t keep;
  if constexpr (NT)
    asm volatile(OMNI_DMA_HEAD
        "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1 nt\n\t"
        OMNI_DMA_TAIL
        : "=&s"(keep) : "s"(sbase), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(lds_dst) : "memory", "scc");
  else
    asm volatile(OMNI_DMA_HEAD
        "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
        OMNI_DMA_TAIL
        : "=&s"(keep) : "s"(sbase), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(lds_dst) : "memory", "scc");
}
// per-group second-level scales and zeros of one k-step: one dword per lane each (256 B per piece)
__device__ __forceinline__ void lds_dma_gp(const void* sb_scales, const void* sb_zeros, uint32_t v0, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(OMNI_DMA_HEAD
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %3, %1\n\t"
      "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %3, %2\n\t"
      OMNI_DMA_TAIL
      : "=&s"(keep) : "s"(sb_scales), "s"(sb_zeros), "v"(v0), "s"(lds_dst) : "memory", "scc");
}
// one piece (16 B per lane, or 4 B per lane with DW) as its own statement: the K loop spreads a chunk's pieces over its MFMAs.
// Three instructions: M0 is declared clobbered instead of saved and restored (nothing else in these kernels uses it), and the
// scalar base / destination come out of SALU arithmetic (no VALU-written SGPR in front of the VMEM instruction: no wait states
// beyond the one between the M0 write and its use).
template <bool NT, bool DW>
__device__ __forceinline__ void lds_dma_piece(const void* sbase, uint32_t v0, uint32_t lds_dst) {
  if constexpr (DW)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %0" ::"s"(sbase), "v"(v0), "s"(lds_dst) : "memory", "m0");
  else if constexpr (NT)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 nt" ::"s"(sbase), "v"(v0), "s"(lds_dst) : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(v0), "s"(lds_dst) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }


#ifdef OMNI_DEBUG_CLOCKS
// timeline probe (tools/midm_timeline.py): per wave of workgroups 0 and gridDim.x - 1, shader-clock stamps
//   [0] entry, [1] prologue requests issued, then per chunk c < 24: [4 + 4c] before the DMA wait, [+1] behind it, [+2] behind the
//   barrier, [+3] behind the chunk's last MFMA issue; [100] K loop done, [101] partials exchanged, [102] stores issued
static __device__ unsigned long long omni_dbg_midm[2 * 8 * 104];

#define MIDM_STAMP(i)                                                                                              \
  do {                                                                                                             \
    if (dbg_on && lane == 0) dbg_t[(i)] = __builtin_readcyclecounter();                                            \
  } while (0)
#else
#define MIDM_STAMP(i) do {} while (0)
#endif

template <int MB, int MODE, bool TO_SLAB, bool NT>
__global__ __launch_bounds__(512, 2) void w4a8_midm_kernel(GemmArgs p) {
  static_assert(MB == 4 || MB == 8, "row tile of 64 or 128 rows (quarters of whole 16-row blocks)");
  constexpr int MT = MB * 16;
  constexpr int NG = 2, NW = 8;
  constexpr int CH = KCHUNK;                               // k per chunk: four 64-k steps, one per K phase
  constexpr int NI = MT / 32;                              // activation pieces per wave and chunk (MT rows / 8 waves / 4 rows)
  constexpr int WL = (MODE == MODE_W8) ? 4 : 2;            // weight pieces per wave and chunk
  constexpr int GP = (MODE == MODE_GRP) ? 1 : 0;           // second-level parameter pieces (one of the pair's two waves: scales, the other: zeros)
  constexpr int OPS = NI + WL + GP;                        // vector memory operations per wave and chunk
  // ring depth: what fits 160 KiB next to the epilogue operands.  128 rows: 3 x (32 KiB tile + 8 x 2 KiB of weights) =
  // 144 KiB (W8A8 rows are twice the bytes: 2 slots); 64 rows: 4 x (16 + 16) = 128 KiB.  D = chunks in flight behind the one
  // being multiplied; vmcnt counts at most 63 operations: (NBUF - 1) x OPS <= 32.
  constexpr int NBUF = MB == 8 ? (MODE == MODE_W8 ? 2 : 3) : (MODE == MODE_W8 ? 3 : 4);
  constexpr int D = NBUF - 1;
  static_assert(D * OPS < 64, "vmcnt is a 6-bit counter");
  constexpr int HB = MB / 2, QB = MB / 4;
  constexpr int ABUF = MT * CH;                            // one activation buffer
  constexpr int WR = (MODE == MODE_W8) ? 4 : 2;            // 16-B reads of a lane per k-step of packed weights
  constexpr int WSLOT = 2 * WL * 1024 + (GP ? 512 : 0);    // one ring slot of one wave PAIR: the two k-steps of its K half
  constexpr int LDS_A = NBUF * ABUF;
  constexpr int LDS_W = (NW / 2) * NBUF * WSLOT;
  constexpr int LDS_RED = NW * QB * 4 * 1024;              // the exchange: every wave parks half of its accumulators
  constexpr int LDS_MAIN = LDS_RED > LDS_A + LDS_W ? LDS_RED : LDS_A + LDS_W;
  constexpr int LDS_EPI = TO_SLAB ? 0 : (64 * NG + MT) * 4;
  static_assert(LDS_MAIN + LDS_EPI <= 160 * 1024, "LDS");
  static_assert(STEPS == 4 && KCHUNK == 256, "one k-step of a 256-k chunk per K phase");
  __shared__ __attribute__((aligned(1024))) uint8_t smem[LDS_MAIN + LDS_EPI];
  uint32_t* const epi_w = reinterpret_cast<uint32_t*>(smem + LDS_MAIN);              // {wscale, w_sz} per channel of the tile
  uint32_t* const epi_a = reinterpret_cast<uint32_t*>(smem + LDS_MAIN) + 64 * NG;    // {ascale, asum} per row of the tile
  // one batch of scalar loads for the prologue's kernel arguments
  asm volatile("" ::"s"(p.A), "s"(p.W), "s"(p.wscales), "s"(p.ascales), "s"(p.wsz), "s"(p.asum), "s"(p.M), "s"(p.N), "s"(p.K), "s"(p.kslice));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave (g, hk, r): channel group g, K HALF hk (k-steps 2 hk, 2 hk + 1 of every chunk), row HALF r (row blocks r * HB ..).
  // (Through version 6 the eight waves were 2 groups x 4 K phases over all rows: 128 accumulator registers per wave and an
  // exchange of 192 KiB through the LDS store path -- 79 B/clk -- in two rounds, 6 100 of a launch's 45 000 cycles; here a
  // wave holds 64, one round moves 64 KiB, and a wave unpacks two k-steps per chunk instead of one: profiles/r05_a.)
  const int g = wave >> 2, hk = (wave >> 1) & 1, r = wave & 1;
  const int s = 2 * hk + r;                                 // the k-step this wave REQUESTS (its pair reads both)
  const int ng = blockIdx.x * NG + g;                       // 64-channel group
  const int m0 = blockIdx.z * MT;
  const int k0 = (int)blockIdx.y * p.kslice;
  const int nchunks = p.kslice / CH;
#ifdef OMNI_DEBUG_CLOCKS
  const bool dbg_on = blockIdx.y == 0 && blockIdx.z == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1);
  unsigned long long* const dbg_t = omni_dbg_midm + ((blockIdx.x == 0 ? 0 : 8) + wave) * 104;
  MIDM_STAMP(0);
#endif

  // ---- request addressing ---------------------------------------------------------------------------------------
  const int lx = (lane >> 3) & 1, lc = lane & 7, le = lane >> 4;
  // Every request is QUAD-COALESCED: four consecutive lanes ask for 64 consecutive bytes in ascending order.  The CU's
  // texture-address unit takes one such quad per clock; a quad whose lanes point at four different 64-B segments (the MFMA
  // operand's own lane map over the packed tile: lanes 64 B apart) or at one segment in permuted order (a full XOR swizzle of
  // the activation row) goes through at a quarter of that -- 16 B/clk per CU, which IS what the first two versions of this
  // kernel ran at (48 KiB per chunk and CU at 128 rows: 1.5 us; profiles/r05_a).  An LDS-DMA image is lane-linear, but the
  // lane that fetches a piece need not be the lane that consumes it: the requests below are laid out for the memory path,
  // and the consumers pick their pieces out of LDS (2-way bank conflicts on those reads: 10 + 2 per 32 MFMAs).
  // weights: instruction j of a k-step = 1 KiB contiguous.  W4: tiles 2 s, 2 s + 1 of tile row 2 ng + j (lane l: bytes
  // 16 l ..); W8: the 16 rows of block j, 64 B each (lane l: row l >> 2, 16-B piece l & 3).  Offsets from p.W, chunk 0.
  uint32_t wvo[MODE == MODE_W8 ? 4 : 2];
  if constexpr (MODE == MODE_W8) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      wvo[j] = (uint32_t)(ng * 64 + j * 16 + (lane >> 2)) * (uint32_t)p.K + (uint32_t)((lane & 3) * 16 + k0 + s * KSTEP);
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      wvo[j] = ((uint32_t)(2 * ng + j) * (uint32_t)(p.K / 32) + (uint32_t)((k0 + s * KSTEP) / 32)) * 512u + (uint32_t)(lane * 16);
  }
  // where this lane's operand pieces landed inside a ring slot.  W4: chunk (n3 = lc, k6 = le) of tile (parity tp) of tile
  // row lx: lx * 1024 + tp * 512 + (lc * 4 + le) * 16; W8: row lane & 15 of block j, piece lane >> 4: j * 1024 + row * 64 + piece * 16
  const uint32_t wrd = MODE == MODE_W8 ? (uint32_t)((lane & 15) * 64 + (lane >> 4) * 16) : (uint32_t)(lx * 1024 + (lc * 4 + le) * 16);
  const uint32_t gvo = (uint32_t)((2 * ng + lx) * 32 + lc * 4);        // per-group parameter column of this lane
  // activations: a piece = 4 rows x 256 B; LDS row m keeps its 16-B piece q at slot q ^ ((m & 3) << 2) -- whole 64-B groups
  // move, the order inside a group stays (a full q ^ (m & 15) is conflict-free for the B reads but permutes the quads).
  // lane -> (row dr of the piece's 4, slot ds)
  uint32_t dvo[NI];
  {
    const int dr = lane >> 4, ds = lane & 15;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int rl = wave * (MT / NW) + i * 4 + dr;                 // row inside the tile
      const int row = (m0 + rl) < p.M ? (m0 + rl) : (p.M - 1);      // rows beyond M re-read the last row (never stored)
      dvo[i] = (uint32_t)row * (uint32_t)p.K + (uint32_t)((ds ^ ((rl & 3) << 2)) << 4);
    }
  }
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)smem);
  const uint32_t wring = __builtin_amdgcn_readfirstlane((uint32_t)(LDS_A + (wave >> 1) * NBUF * WSLOT));   // this pair's ring, from smem
  // all requests of chunk c into buffer / slot b
  // request number k (static) of chunk c into buffer / slot b: activation pieces first, then weights, then parameters
  auto issue_piece = [&](auto k_tag, int c, int b) {
    constexpr int k = decltype(k_tag)::value;
    if constexpr (k < NI) {
      const uint8_t* sa = reinterpret_cast<const uint8_t*>(p.A) + (size_t)k0 + (size_t)c * CH;
      const uint32_t adst = lds_base + (uint32_t)b * ABUF + (uint32_t)wave * (MT / NW) * 256 + k * 1024;
      lds_dma_piece<false, false>(sa, dvo[k], adst);
    } else if constexpr (k < NI + WL) {
      constexpr int j = k - NI;
      const uint32_t wdst = lds_base + wring + (uint32_t)b * WSLOT + (uint32_t)r * (WL * 1024) + j * 1024;
      if constexpr (MODE == MODE_W8) lds_dma_piece<NT, false>(p.W + (size_t)c * CH, wvo[j], wdst);
      else lds_dma_piece<NT, false>(p.W + (size_t)c * (CH / 32) * 512, wvo[j], wdst);
    } else if constexpr (k < OPS) {
      const size_t grow = (size_t)(k0 / 128 + 2 * c + hk) * p.N;      // 128-k group of (chunk c, K half hk): both k-steps
      const uint32_t gdst = lds_base + wring + (uint32_t)b * WSLOT + 2 * WL * 1024 + (uint32_t)r * 256;      // r = 0: scales, 1: zeros
      lds_dma_piece<false, true>((r ? p.s2z : p.s2s) + grow, gvo, gdst);
    }
  };
  auto issue = [&](int c, int b) {      // all of them (prologue)
    issue_piece(IntTag<0>{}, c, b); issue_piece(IntTag<1>{}, c, b); issue_piece(IntTag<2>{}, c, b); issue_piece(IntTag<3>{}, c, b);
    issue_piece(IntTag<4>{}, c, b); issue_piece(IntTag<5>{}, c, b); issue_piece(IntTag<6>{}, c, b); issue_piece(IntTag<7>{}, c, b);
    static_assert(OPS <= 8, "pieces are enumerated");
  };
  // B operand of this wave's first row block at k-step 2 hk (per lane); k-step 2 hk + 1: ^ 64; row block mb adds an immediate
  const uint32_t boff = (uint32_t)(r * HB * 16 * 256) + (uint32_t)(lane & 15) * 256 +
                        ((uint32_t)(((lane >> 4) ^ ((lane & 3) << 2)) << 4) ^ (uint32_t)(hk << 7));

  v4i acc[HB][4];
#pragma unroll
  for (int mb = 0; mb < HB; ++mb)
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) acc[mb][ab] = (v4i){0, 0, 0, 0};

  // ---- prologue: the first D chunks are requested; epilogue operands -> LDS (published by the first chunk's barrier) ----
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nchunks) issue(d, d);
  MIDM_STAMP(1);
  if constexpr (!TO_SLAB) {
    if (tid < 64 * NG) {
      const int n = blockIdx.x * 64 * NG + tid;
      const uint32_t sw = __builtin_bit_cast(uint16_t, p.wscales[n]);
      uint32_t sz = 0;
      if constexpr (MODE == MODE_CHN) sz = __builtin_bit_cast(uint16_t, p.wsz[n]);
      epi_w[tid] = sw | (sz << 16);
    } else if (tid - 64 * NG < MT) {
      const int i = tid - 64 * NG;
      const int m = (m0 + i) < p.M ? (m0 + i) : (p.M - 1);
      const uint32_t sa = __builtin_bit_cast(uint16_t, p.ascales[m]);
      uint32_t as = 0;
      if constexpr (MODE == MODE_CHN) as = __builtin_bit_cast(uint16_t, p.asum[m]);
      epi_a[i] = sa | (as << 16);
    }
  }

  // Software pipeline across the barrier: the MFMAs of a chunk's LAST operand (k-step 2 hk + 1, operand 3: HB of them) run at
  // the HEAD of the next chunk, behind its barrier and its first LDS reads -- there the weight reads, the lane transposition
  // and the first masks are in flight and no MFMA of the new chunk can issue yet (both waves of the SIMD).  Their operands stay
  // in registers (`wa_def`, `bf_def`); zero operands in front of the first chunk: those MFMAs add nothing.
  v4i wa[4], wa_def, bf_def[HB];
  wa_def = (v4i){0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < HB; ++j) bf_def[j] = (v4i){0, 0, 0, 0};
  auto deferred = [&]() {
#pragma unroll
    for (int mb = 0; mb < HB; ++mb)
      if (!(OMNI_MIDM_ABLATE & 16) || mb == 0) acc[mb][3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa_def, bf_def[mb], acc[mb][3], 0, 0, 0);
  };

  // ---- one chunk: this wave's two k-steps of it, out of buffer / slot B (static).  STEADY: chunk c + D exists ----------------
  auto body = [&](int c, auto buf_tag, auto steady_tag) {
    constexpr int B = decltype(buf_tag)::value;
    constexpr bool STEADY = decltype(steady_tag)::value;
    // my pieces of chunk c have landed: behind them the requests of the chunks c + 1 .. min(c + D - 1, last)
    if (c < 24) MIDM_STAMP(4 + 4 * c);
    if constexpr (STEADY) {
      vm_wait<(D - 1) * OPS>();
    } else {
      const int after = nchunks - 1 - c < D - 1 ? nchunks - 1 - c : D - 1;
      if (after <= 0) vm_wait<0>();
      else if (after == 1) vm_wait<1 * OPS>();
      else vm_wait<(D > 2 ? 2 : 1) * OPS>();
      static_assert(D <= 3, "tail waits are enumerated");
    }
    if (c < 24) MIDM_STAMP(5 + 4 * c);
    if (!(OMNI_MIDM_ABLATE & 8)) __syncthreads();          // chunk c is visible in buffer B; everybody is done reading buffer (B + D) % NBUF (chunk c - 1)
    if (c < 24) MIDM_STAMP(6 + 4 * c);
    const bool more = (OMNI_MIDM_ABLATE & 1) ? false : (STEADY || c + D < nchunks);      // chunk c + D is requested during this chunk
    const uint8_t* abuf = smem + B * ABUF;
    const uint8_t* psl = smem + wring + B * WSLOT;
    // The chunk runs as STAGES fenced against each other (sched_barrier): HB MFMAs of one operand, the unpack of the next
    // operand under them, the LDS reads a later stage needs, and ONE request of chunk c + D.  Issued in a burst behind the
    // barrier, the 48 requests of a workgroup queue up in the CU's address unit (16 clocks each) while every wave waits to get
    // its next one accepted and the matrix pipe idles (tools/midm_timeline.py, profiles/r05_a).
    constexpr int NSTAGE = 9;          // the deferred operand, then 2 k-steps x 4 operand stages
    // request k goes out in stage k * NSTAGE / OPS (spread evenly; several per stage where there are more requests than stages)
    auto stage_requests = [&](auto st_tag) {
      constexpr int st = decltype(st_tag)::value;
      if (more) {
        if constexpr (0 * NSTAGE / OPS == st && 0 < OPS) issue_piece(IntTag<0>{}, c + D, (B + D) % NBUF);
        if constexpr (1 * NSTAGE / OPS == st && 1 < OPS) issue_piece(IntTag<1>{}, c + D, (B + D) % NBUF);
        if constexpr (2 * NSTAGE / OPS == st && 2 < OPS) issue_piece(IntTag<2>{}, c + D, (B + D) % NBUF);
        if constexpr (3 * NSTAGE / OPS == st && 3 < OPS) issue_piece(IntTag<3>{}, c + D, (B + D) % NBUF);
        if constexpr (4 * NSTAGE / OPS == st && 4 < OPS) issue_piece(IntTag<4>{}, c + D, (B + D) % NBUF);
        if constexpr (5 * NSTAGE / OPS == st && 5 < OPS) issue_piece(IntTag<5>{}, c + D, (B + D) % NBUF);
        if constexpr (6 * NSTAGE / OPS == st && 6 < OPS) issue_piece(IntTag<6>{}, c + D, (B + D) % NBUF);
        if constexpr (7 * NSTAGE / OPS == st && 7 < OPS) issue_piece(IntTag<7>{}, c + D, (B + D) % NBUF);
      }
    };
    v4i bf[2][HB];
    uint32_t d[2][4];
    uint32_t sc4 = 0, zr4 = 0;
    uint4 wraw[2][WR];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < WR; ++j) wraw[kk][j] = *reinterpret_cast<const uint4*>(psl + kk * (WL * 1024) + wrd + j * (MODE == MODE_W8 ? 1024 : 512));
    if constexpr (MODE == MODE_GRP) {
      sc4 = *reinterpret_cast<const uint32_t*>(psl + 2 * WL * 1024 + lane * 4);
      zr4 = *reinterpret_cast<const uint32_t*>(psl + 2 * WL * 1024 + 256 + lane * 4);
    }
#pragma unroll
    for (int mb = 0; mb < HB; ++mb) bf[0][mb] = *reinterpret_cast<const v4i*>(abuf + boff + ((OMNI_MIDM_ABLATE & 2) ? 0 : mb) * 16 * 256);
    deferred();               // (chunk c - 1's last operand: stage 0)
    stage_requests(IntTag<0>{});
    auto transpose = [&](auto kk_tag) {
      constexpr int kk = decltype(kk_tag)::value;
      if constexpr (MODE != MODE_W8) {
        // dwords of a 16-B piece: x = (k5 = 0, n2 = 0) y = (0, 1) z = (1, 0) w = (1, 1); d[n2][(tile parity, k5)]
        const uint32_t dd[2][4] = {{wraw[kk][0].x, wraw[kk][0].z, wraw[kk][1].x, wraw[kk][1].z},
                                   {wraw[kk][0].y, wraw[kk][0].w, wraw[kk][1].y, wraw[kk][1].w}};
        // register index (tile parity, k5) <-> 16-lane row k6: afterwards d[b][q] = k6 = q of (parity, k5) = lane >> 4,
        // i.e. 16 consecutive k per lane -- what a DMA-written activation row offers (w4a8_gemm_exact_kernel)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const auto s02 = __builtin_amdgcn_permlane32_swap(dd[b][0], dd[b][2], false, false);
          const auto s13 = __builtin_amdgcn_permlane32_swap(dd[b][1], dd[b][3], false, false);
          const auto s01 = __builtin_amdgcn_permlane16_swap((uint32_t)s02[0], (uint32_t)s13[0], false, false);
          const auto s23 = __builtin_amdgcn_permlane16_swap((uint32_t)s02[1], (uint32_t)s13[1], false, false);
          d[b][0] = (uint32_t)s01[0]; d[b][1] = (uint32_t)s01[1]; d[b][2] = (uint32_t)s23[0]; d[b][3] = (uint32_t)s23[1];
        }
      }
    };
    auto operand = [&](auto kk_tag, auto ab_tag) {      // MFMA A operand ab = a * 2 + b of k-step 2 hk + kk
      constexpr int kk = decltype(kk_tag)::value;
      constexpr int ab = decltype(ab_tag)::value;
      if constexpr (MODE == MODE_W8) {
        wa[ab] = (v4i){(int)wraw[kk][ab].x, (int)wraw[kk][ab].y, (int)wraw[kk][ab].z, (int)wraw[kk][ab].w};
      } else {
        constexpr int a = ab >> 1, b = ab & 1;
        uint32_t u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q] = (OMNI_MIDM_ABLATE & 4) ? d[b][q] : ((d[b][q] >> (4 * a)) & 0x0F0F0F0Fu);
        if constexpr (MODE == MODE_GRP) {
          const uint32_t sc = (sc4 >> (8 * ab)) & 0xFFu;
#pragma unroll
          for (int q = 0; q < 4; ++q) u[q] = u[q] * sc;
          vadd4_zbyte_x4(u, zr4, ab);
        }
        wa[ab] = (v4i){(int)u[0], (int)u[1], (int)u[2], (int)u[3]};
      }
    };
    auto operand_stage = [&](auto kk_tag, auto ab_tag) {
      constexpr int kk = decltype(kk_tag)::value;
      constexpr int ab = decltype(ab_tag)::value;
      operand(kk_tag, ab_tag);
      if constexpr (kk == 1 && ab == 3) {      // the chunk's last operand: its MFMAs wait behind the next barrier
        wa_def = wa[3];
#pragma unroll
        for (int mb = 0; mb < HB; ++mb) bf_def[mb] = bf[1][mb];
      } else {
#pragma unroll
        for (int mb = 0; mb < HB; ++mb)
          if (!(OMNI_MIDM_ABLATE & 16) || mb == 0) acc[mb][ab] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[ab], bf[kk][mb], acc[mb][ab], 0, 0, 0);
      }
      // the second k-step's B operands: one per stage of the first
      if constexpr (kk == 0 && ab < HB)
        bf[1][ab] = *reinterpret_cast<const v4i*>(abuf + (boff ^ 64u) + ((OMNI_MIDM_ABLATE & 2) ? 0 : ab) * 16 * 256);
      stage_requests(IntTag<1 + 4 * kk + ab>{});
      __builtin_amdgcn_sched_barrier(0);
    };
    transpose(IntTag<0>{});
    __builtin_amdgcn_sched_barrier(0);
    operand_stage(IntTag<0>{}, IntTag<0>{});
    operand_stage(IntTag<0>{}, IntTag<1>{});
    operand_stage(IntTag<0>{}, IntTag<2>{});
    operand_stage(IntTag<0>{}, IntTag<3>{});
    transpose(IntTag<1>{});
    __builtin_amdgcn_sched_barrier(0);
    operand_stage(IntTag<1>{}, IntTag<0>{});
    operand_stage(IntTag<1>{}, IntTag<1>{});
    operand_stage(IntTag<1>{}, IntTag<2>{});
    operand_stage(IntTag<1>{}, IntTag<3>{});
    static_assert(HB <= 4, "the second k-step's B reads ride on the first one's four stages");
#ifdef OMNI_DEBUG_CLOCKS
    __builtin_amdgcn_sched_barrier(0);
    if (c < 24) MIDM_STAMP(7 + 4 * c);
#endif
  };
  {
    int c = 0;
    for (; c + (NBUF - 1) + D < nchunks; c += NBUF) {      // whole ring rounds whose every chunk has a chunk c + D behind it
      body(c, IntTag<0>{}, BoolTag<true>{});
      body(c + 1, IntTag<1 % NBUF>{}, BoolTag<true>{});
      if constexpr (NBUF > 2) body(c + 2, IntTag<2 % NBUF>{}, BoolTag<true>{});
      if constexpr (NBUF > 3) body(c + 3, IntTag<3 % NBUF>{}, BoolTag<true>{});
      static_assert(NBUF >= 2 && NBUF <= 4, "ring rounds are unrolled by hand");
    }
    for (; c < nchunks; c += NBUF) {                        // the last chunks: conditions on workgroup-uniform values
      body(c, IntTag<0>{}, BoolTag<false>{});
      if (c + 1 < nchunks) body(c + 1, IntTag<1 % NBUF>{}, BoolTag<false>{});
      if (NBUF > 2 && c + 2 < nchunks) body(c + 2, IntTag<2 % NBUF>{}, BoolTag<false>{});
      if (NBUF > 3 && c + 3 < nchunks) body(c + 3, IntTag<3 % NBUF>{}, BoolTag<false>{});
    }
  }
  deferred();                 // the last chunk's
  __builtin_amdgcn_sched_barrier(0);
  MIDM_STAMP(100);

  // ---- the two K halves of a (group, row half) meet in LDS (static accumulator indices only) ----------------------------
  // K half 0 keeps the low QB row blocks of its row half and parks the high ones, K half 1 the other way round; partner = wave ^ 2.
  // Afterwards wave (g, hk, r) holds the finished accumulators of its row blocks hk * QB .. + QB - 1 (tile row blocks r * HB + ..).
  __syncthreads();            // the activation buffers are free
  v4i* const red = reinterpret_cast<v4i*>(smem);
  {
    v4i* const mine = red + (size_t)wave * QB * 4 * 64 + lane;
    const v4i* const theirs = red + (size_t)(wave ^ 2) * QB * 4 * 64 + lane;
    if (hk == 0) {
#pragma unroll
      for (int j = 0; j < QB; ++j)
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) mine[(j * 4 + ab) * 64] = acc[QB + j][ab];
    } else {
#pragma unroll
      for (int j = 0; j < QB; ++j)
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) mine[(j * 4 + ab) * 64] = acc[j][ab];
    }
    __syncthreads();
    if (hk == 0) {
#pragma unroll
      for (int j = 0; j < QB; ++j)
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) acc[j][ab] += theirs[(j * 4 + ab) * 64];
    } else {
#pragma unroll
      for (int j = 0; j < QB; ++j)
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) acc[QB + j][ab] += theirs[(j * 4 + ab) * 64];
    }
  }

  MIDM_STAMP(101);
  // ---- write back: this wave's row blocks hk * QB .. + QB - 1 (tile row blocks r * HB + ..) of group g --------------------
  // D layout (16x16): col = lane & 15 -> row m of the block, row = (lane >> 4) * 4 + r -> channel slot i.
  // W4: channel = ng * 64 + (i >> 3) * 32 + ab * 8 + (i & 7) (4 consecutive channels per lane); W8: ng * 64 + ab * 16 + i.
  const int mcol = lane & 15;
  const int i0 = (lane >> 4) * 4;
  const bool out16 = !TO_SLAB && p.tile_linear != 0;      // (host: output rows start 16-B aligned)
  auto finish = [&](auto first_tag) {
    constexpr int FIRST = decltype(first_tag)::value;
#pragma unroll
    for (int j = 0; j < QB; ++j) {
      const int mb = FIRST + j;               // accumulator index (static)
      const int tmb = r * HB + mb;            // row block of the tile
      const int m = m0 + tmb * 16 + mcol;
      float sa = 0.f, as = 0.f;
      if constexpr (!TO_SLAB) {
        const uint32_t av = epi_a[tmb * 16 + mcol];
        sa = (float)__builtin_bit_cast(half_t, (uint16_t)(av & 0xFFFFu));
        as = (float)__builtin_bit_cast(half_t, (uint16_t)(av >> 16));
      }
      auto fp16x4 = [&](int ab) -> uint2 {      // the four channels this lane holds of operand block ab, finished to fp16
        int nl;     // channel inside the workgroup's tile
        if constexpr (MODE == MODE_W8) nl = g * 64 + ab * 16 + i0;
        else nl = g * 64 + (i0 >> 3) * 32 + ab * 8 + (i0 & 7);
        const uint4 w4 = *reinterpret_cast<const uint4*>(&epi_w[nl]);      // {wscale, w_sz} x 4 channels
        const uint32_t wv[4] = {w4.x, w4.y, w4.z, w4.w};
        const v4i a4 = acc[mb][ab];
        half_t o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          o[r] = epilogue<MODE>(a4[r], (float)__builtin_bit_cast(half_t, (uint16_t)(wv[r] & 0xFFFFu)), sa,
                                (float)__builtin_bit_cast(half_t, (uint16_t)(wv[r] >> 16)), as);
        return *reinterpret_cast<const uint2*>(o);
      };
      if constexpr (TO_SLAB) {
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
          int nl;
          if constexpr (MODE == MODE_W8) nl = g * 64 + ab * 16 + i0;
          else nl = g * 64 + (i0 >> 3) * 32 + ab * 8 + (i0 & 7);
          if (m < p.M)
            *reinterpret_cast<v4i*>(p.slab + ((size_t)blockIdx.y * p.M + m) * p.N + blockIdx.x * 64 * NG + nl) = acc[mb][ab];
        }
      } else if (out16) {
        // lane pairs (l, l ^ 16) exchange halves so that every lane stores 16 B = 8 consecutive channels
        // (w4a8_gemm_exact_kernel's write-back): 4 store instructions per wave and two row blocks instead of 8
        const int odd = (lane >> 4) & 1;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const uint2 x = fp16x4(2 * pr), y = fp16x4(2 * pr + 1);
          const auto lo = __builtin_amdgcn_permlane16_swap(x.x, y.x, false, false);
          const auto hi = __builtin_amdgcn_permlane16_swap(x.y, y.y, false, false);
          int n8;
          if constexpr (MODE == MODE_W8) n8 = g * 64 + (2 * pr + odd) * 16 + (lane >> 5) * 8;
          else n8 = g * 64 + (lane >> 5) * 32 + (2 * pr + odd) * 8;
          if (m < p.M)
            *reinterpret_cast<uint4*>(p.out + (size_t)m * p.out_stride + blockIdx.x * 64 * NG + n8) =
                make_uint4((uint32_t)lo[0], (uint32_t)hi[0], (uint32_t)lo[1], (uint32_t)hi[1]);
        }
      } else {
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
          int nl;
          if constexpr (MODE == MODE_W8) nl = g * 64 + ab * 16 + i0;
          else nl = g * 64 + (i0 >> 3) * 32 + ab * 8 + (i0 & 7);
          const uint2 o = fp16x4(ab);
          if (m < p.M) *reinterpret_cast<uint2*>(p.out + (size_t)m * p.out_stride + blockIdx.x * 64 * NG + nl) = o;
        }
      }
    }
  };
  if (hk == 0) finish(IntTag<0>{});
  else finish(IntTag<QB>{});
  MIDM_STAMP(102);
}

}  // namespace omni
