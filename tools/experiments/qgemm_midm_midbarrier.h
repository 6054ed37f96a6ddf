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
//   [0] entry, [1] prologue requests issued, then per chunk c < 24: [4 + 4c] top, [+1] behind the first half's MFMA issue,
//   [+2] behind the DMA wait and the barrier, [+3] behind the second half; [100] K loop done, [101] partials exchanged,
//   [102] stores issued
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
  constexpr int GP = (MODE == MODE_GRP) ? 2 : 0;           // second-level parameter pieces
  constexpr int OPS = NI + WL + GP;                        // vector memory operations per wave and chunk
  // ring depth: what fits 160 KiB next to the epilogue operands.  128 rows: 3 x (32 KiB tile + 8 x 2 KiB of weights) =
  // 144 KiB (W8A8 rows are twice the bytes: 2 slots); 64 rows: 4 x (16 + 16) = 128 KiB.  D = chunks in flight behind the one
  // being multiplied; vmcnt counts at most 63 operations: (NBUF - 1) x OPS <= 32.
  constexpr int NBUF = (MB == 8 && MODE == MODE_W8) ? 2 : 3;
  static_assert(NBUF * OPS < 64, "vmcnt is a 6-bit counter");
  constexpr int HB = MB / 2, QB = MB / 4;
  constexpr int ABUF = MT * CH;                            // one activation buffer
  constexpr int WSLOT = WL * 1024 + (GP ? 512 : 0);        // one ring slot of one wave
  constexpr int LDS_A = NBUF * ABUF;
  constexpr int LDS_W = NW * NBUF * WSLOT;
  constexpr int LDS_RED = NW * HB * 4 * 1024;              // first exchange round: every wave parks half of its accumulators
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
  const int g = wave >> 2, s = wave & 3;                    // channel group inside the tile, K phase
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
  const uint32_t wring = __builtin_amdgcn_readfirstlane((uint32_t)(LDS_A + wave * NBUF * WSLOT));   // this wave's ring, from smem
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
      const uint32_t wdst = lds_base + wring + (uint32_t)b * WSLOT + j * 1024;
      if constexpr (MODE == MODE_W8) lds_dma_piece<NT, false>(p.W + (size_t)c * CH, wvo[j], wdst);
      else lds_dma_piece<NT, false>(p.W + (size_t)c * (CH / 32) * 512, wvo[j], wdst);
    } else if constexpr (k < OPS) {
      constexpr int j = k - NI - WL;      // 0: scales, 1: zeros
      const size_t grow = (size_t)(k0 / 128 + 2 * c + (s >> 1)) * p.N;      // 128-k group of (chunk c, phase s)
      const uint32_t gdst = lds_base + wring + (uint32_t)b * WSLOT + WL * 1024 + j * 256;
      lds_dma_piece<false, true>((j ? p.s2z : p.s2s) + grow, gvo, gdst);
    }
  };
  auto issue = [&](int c, int b) {      // all of them (prologue)
    issue_piece(IntTag<0>{}, c, b); issue_piece(IntTag<1>{}, c, b); issue_piece(IntTag<2>{}, c, b); issue_piece(IntTag<3>{}, c, b);
    issue_piece(IntTag<4>{}, c, b); issue_piece(IntTag<5>{}, c, b); issue_piece(IntTag<6>{}, c, b); issue_piece(IntTag<7>{}, c, b);
    static_assert(OPS <= 8, "pieces are enumerated");
  };
  // B operand of row block 0 at this wave's k-step (per lane); row block mb adds an immediate
  const uint32_t boff = (uint32_t)(lane & 15) * 256 + ((uint32_t)(((lane >> 4) ^ ((lane & 3) << 2)) << 4) ^ (uint32_t)(s << 6));

  v4i acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) acc[mb][ab] = (v4i){0, 0, 0, 0};

  // ---- K loop, software-pipelined around a MID-CHUNK barrier ---------------------------------------------------
  // An in-order wave cannot multiply while it waits for its own LDS reads, lane transposition and masks, and the two
  // waves of a SIMD pass every barrier together: with the barrier at the TOP of a chunk both did that preparation with the
  // matrix pipe idle (chunks of 2050 cycles against 1024 of MFMA per SIMD; timing ablations: without any memory request
  // 20.8 us, with a quarter of the MFMAs 19.5 us, against 22.1 -- a latency chain, not a resource; profiles/r05_a §2).
  // So the barrier that publishes chunk c + 1 sits in the MIDDLE of chunk c's MFMAs:
  //   P0 (rows 0 .. HALF - 1 of chunk c, row by row): all remaining B operands of chunk c are read (nothing of chunk c is
  //       read behind the barrier: its buffer is recycled there) and the weight-side requests of chunk c + NBUF go out;
  //   wait for this wave's pieces of chunk c + 1, barrier;
  //   P1 (rows HALF .. MB - 1, operand by operand): the reads of chunk c + 1's packed weights and first B rows are issued,
  //       and while the MFMAs of operands 0, 1 run the data arrives; operand ab of chunk c + 1 is unpacked into wa[ab] right
  //       behind the last MFMA that reads the old wa[ab]; the activation requests of chunk c + NBUF go out (buffer c % NBUF
  //       is free behind the barrier).
  // Every request is issued NBUF chunks ahead and has two chunks (NBUF = 3) to land.
  constexpr int HALF = MB / 2, PF = 2;
  static_assert(HALF >= PF, "two prefetched B rows");
  constexpr int WP = WL + GP;                                   // weight-side requests per chunk
  constexpr int VM_STEADY = (NBUF - 2) * OPS + WP;              // requests issued behind chunk c + 1's by the middle of chunk c
  v4i wa[4], bf[MB];
  uint32_t d[2][4];
  uint32_t sc4 = 0, zr4 = 0;
  uint4 wraw[MODE == MODE_W8 ? 4 : 2];
  auto read_weights = [&](int b) {       // this wave's packed k-step out of ring slot b (+ its second-level parameters)
    const uint8_t* wsl = smem + wring + b * WSLOT + wrd;
#pragma unroll
    for (int j = 0; j < (MODE == MODE_W8 ? 4 : 2); ++j) wraw[j] = *reinterpret_cast<const uint4*>(wsl + j * (MODE == MODE_W8 ? 1024 : 512));
    if constexpr (MODE == MODE_GRP) {
      sc4 = *reinterpret_cast<const uint32_t*>(smem + wring + b * WSLOT + WL * 1024 + lane * 4);
      zr4 = *reinterpret_cast<const uint32_t*>(smem + wring + b * WSLOT + WL * 1024 + 256 + lane * 4);
    }
  };
  auto read_b = [&](int b, int mb) -> v4i { return *reinterpret_cast<const v4i*>(smem + b * ABUF + boff + mb * 16 * 256); };
  auto transpose = [&]() {
    if constexpr (MODE != MODE_W8) {
      // dwords of a 16-B piece: x = (k5 = 0, n2 = 0) y = (0, 1) z = (1, 0) w = (1, 1); d[n2][(tile parity, k5)]
      const uint32_t dd[2][4] = {{wraw[0].x, wraw[0].z, wraw[1].x, wraw[1].z}, {wraw[0].y, wraw[0].w, wraw[1].y, wraw[1].w}};
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
  auto operand = [&](auto ab_tag) {      // MFMA A operand ab = a * 2 + b of the k-step in `d` / `wraw`
    constexpr int ab = decltype(ab_tag)::value;
    if constexpr (MODE == MODE_W8) {
      wa[ab] = (v4i){(int)wraw[ab].x, (int)wraw[ab].y, (int)wraw[ab].z, (int)wraw[ab].w};
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

  // ---- prologue ------------------------------------------------------------------------------------------------
  // Chunk 0 of every wave first, a barrier, then the other NBUF - 1 chunks: the address unit serves the oldest wave first, and
  // with all of a wave's prologue requests in one run the last waves' chunk-0 pieces queued behind 2 x 48 KiB of the first
  // waves' later chunks.  The epilogue operands are plain loads right behind chunk 0's requests (in-order returns: waiting
  // for chunk 0 waits for them too, and nothing else); they stay in two registers until the K loop is over -- a use any
  // earlier would be waited for with vmcnt(0), i.e. behind every request in flight (the compiler does not see the asm's).
  if (nchunks > 0) issue(0, 0);
  uint32_t epi_lo = 0, epi_hi = 0;
  if constexpr (!TO_SLAB) {
    if (tid < 64 * NG) {
      const int n = blockIdx.x * 64 * NG + tid;
      epi_lo = __builtin_bit_cast(uint16_t, p.wscales[n]);
      if constexpr (MODE == MODE_CHN) epi_hi = __builtin_bit_cast(uint16_t, p.wsz[n]);
    } else if (tid - 64 * NG < MT) {
      const int i = tid - 64 * NG;
      const int m = (m0 + i) < p.M ? (m0 + i) : (p.M - 1);
      epi_lo = __builtin_bit_cast(uint16_t, p.ascales[m]);
      if constexpr (MODE == MODE_CHN) epi_hi = __builtin_bit_cast(uint16_t, p.asum[m]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int c0 = 1; c0 < NBUF; ++c0)
    if (c0 < nchunks) issue(c0, c0);
  MIDM_STAMP(1);
  // chunk 0: its pieces (behind them the other prologue chunks'), the barrier, its operands
  if (nchunks >= NBUF) vm_wait<(NBUF - 1) * OPS>();
  else vm_wait<0>();
  __syncthreads();
  read_weights(0);
#pragma unroll
  for (int mb = 0; mb < PF; ++mb) bf[mb] = read_b(0, mb);
  transpose();
  operand(IntTag<0>{}); operand(IntTag<1>{}); operand(IntTag<2>{}); operand(IntTag<3>{});

  // ---- one chunk out of buffer / slot B (static).  STEADY: the chunks c + 1 and c + NBUF exist ---------------------------
  auto body = [&](int c, auto buf_tag, auto steady_tag) {
    constexpr int B = decltype(buf_tag)::value;
    constexpr int BN = (B + 1) % NBUF;                       // buffer / slot of chunk c + 1
    constexpr bool STEADY = decltype(steady_tag)::value;
    const bool has_next = STEADY || c + 1 < nchunks;
    const bool req = (OMNI_MIDM_ABLATE & 1) ? false : (STEADY || c + NBUF < nchunks);
    if (c < 24) MIDM_STAMP(4 + 4 * c);
    // ---- P0: rows 0 .. HALF - 1, row by row; the rest of chunk c's B operands; weight-side requests of chunk c + NBUF ----
    auto p0_stage = [&](auto i_tag) {
      constexpr int i = decltype(i_tag)::value;
      if constexpr (i < HALF) {
#pragma unroll
        for (int ab = 0; ab < ((OMNI_MIDM_ABLATE & 16) ? 1 : 4); ++ab)
          acc[i][ab] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[ab], bf[i], acc[i][ab], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);      // (reads hoisted above the first MFMA of a loop round would be waited for at once)
        if constexpr (PF + i < HALF) bf[PF + i] = (OMNI_MIDM_ABLATE & 2) ? bf[0] : read_b(B, PF + i);
        bf[HALF + i] = (OMNI_MIDM_ABLATE & 2) ? bf[0] : read_b(B, HALF + i);
        if (req) {      // weight-side request k (k = NI .. OPS - 1) goes out in stage (k - NI) * HALF / WP
          if constexpr ((0 * HALF) / WP == i && 0 < WP) issue_piece(IntTag<NI + 0>{}, c + NBUF, B);
          if constexpr ((1 * HALF) / WP == i && 1 < WP) issue_piece(IntTag<NI + 1>{}, c + NBUF, B);
          if constexpr ((2 * HALF) / WP == i && 2 < WP) issue_piece(IntTag<NI + 2>{}, c + NBUF, B);
          if constexpr ((3 * HALF) / WP == i && 3 < WP) issue_piece(IntTag<NI + 3>{}, c + NBUF, B);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    p0_stage(IntTag<0>{}); p0_stage(IntTag<1>{}); p0_stage(IntTag<2>{}); p0_stage(IntTag<3>{});
    static_assert(HALF <= 4 && WP <= 4, "stages / requests are enumerated");
    if (c < 24) MIDM_STAMP(5 + 4 * c);
    // ---- my pieces of chunk c + 1 have landed; barrier: chunk c + 1 is visible, nobody reads chunk c's buffer any more ----
    if (has_next) {
      // requests issued behind chunk c + 1's: all of chunk c + 2 .. c + NBUF - 1, the weight side of chunk c + NBUF
      if (STEADY || c + NBUF < nchunks) vm_wait<VM_STEADY>();
      else if (NBUF == 3 && c + 2 < nchunks) vm_wait<(NBUF - 2) * OPS>();
      else vm_wait<0>();
      if (!(OMNI_MIDM_ABLATE & 8)) __syncthreads();
    }
    if (c < 24) MIDM_STAMP(6 + 4 * c);
    // ---- P1: rows HALF .. MB - 1, operand by operand; chunk c + 1's operands; activation requests of chunk c + NBUF ----
    v4i nbf[PF];
    if (has_next) {
      read_weights(BN);
#pragma unroll
      for (int mb = 0; mb < PF; ++mb) nbf[mb] = read_b(BN, mb);
    }
    auto p1_mfma = [&](auto ab_tag) {
      constexpr int ab = decltype(ab_tag)::value;
      if (!(OMNI_MIDM_ABLATE & 16) || ab == 0) {
#pragma unroll
        for (int mb = HALF; mb < MB; ++mb) acc[mb][ab] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[ab], bf[mb], acc[mb][ab], 0, 0, 0);
      }
    };
    auto p1_requests = [&](auto st_tag) {      // activation request k (k < NI) goes out in stage k * 4 / NI
      constexpr int st = decltype(st_tag)::value;
      if (req) {
        if constexpr ((0 * 4) / NI == st && 0 < NI) issue_piece(IntTag<0>{}, c + NBUF, B);
        if constexpr ((1 * 4) / NI == st && 1 < NI) issue_piece(IntTag<1>{}, c + NBUF, B);
        if constexpr ((2 * 4) / NI == st && 2 < NI) issue_piece(IntTag<2>{}, c + NBUF, B);
        if constexpr ((3 * 4) / NI == st && 3 < NI) issue_piece(IntTag<3>{}, c + NBUF, B);
      }
    };
    static_assert(NI <= 4, "activation requests are enumerated");
    __builtin_amdgcn_sched_barrier(0);
    p1_mfma(IntTag<0>{}); p1_requests(IntTag<0>{});
    __builtin_amdgcn_sched_barrier(0);
    p1_mfma(IntTag<1>{}); p1_requests(IntTag<1>{});
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) transpose();                 // (the packed weights of chunk c + 1 have had two MFMA groups to arrive)
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) operand(IntTag<0>{});        // wa[0], wa[1] are free: their last MFMAs of chunk c are issued
    p1_mfma(IntTag<2>{}); p1_requests(IntTag<2>{});
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) operand(IntTag<1>{});
    p1_mfma(IntTag<3>{}); p1_requests(IntTag<3>{});
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) {
      operand(IntTag<2>{}); operand(IntTag<3>{});
#pragma unroll
      for (int mb = 0; mb < PF; ++mb) bf[mb] = nbf[mb];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (c < 24) MIDM_STAMP(7 + 4 * c);
  };
  {
    int c = 0;
    for (; c + (NBUF - 1) + NBUF < nchunks; c += NBUF) {      // whole ring rounds whose every chunk has a chunk c + NBUF behind it
      body(c, IntTag<0>{}, BoolTag<true>{});
      body(c + 1, IntTag<1 % NBUF>{}, BoolTag<true>{});
      if constexpr (NBUF > 2) body(c + 2, IntTag<2 % NBUF>{}, BoolTag<true>{});
      static_assert(NBUF == 2 || NBUF == 3, "ring rounds are unrolled by hand");
    }
    for (; c < nchunks; c += NBUF) {                          // the last chunks: conditions on workgroup-uniform values
      body(c, IntTag<0>{}, BoolTag<false>{});
      if (c + 1 < nchunks) body(c + 1, IntTag<1 % NBUF>{}, BoolTag<false>{});
      if (NBUF > 2 && c + 2 < nchunks) body(c + 2, IntTag<2 % NBUF>{}, BoolTag<false>{});
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  MIDM_STAMP(100);

  // ---- the four K phases of a group meet in LDS (static accumulator indices only) ------------------------------
  // round 1: phases {0,1} keep the low half of the row blocks and park the high half, phases {2,3} the other way round;
  //          partner = phase ^ 2.  round 2: inside {0,1} and {2,3} the same with quarters; partner = phase ^ 1.
  // Afterwards wave (g, s) holds the finished accumulators of row blocks s * QB .. s * QB + QB - 1.
  asm volatile("" : "+v"(epi_lo), "+v"(epi_hi));      // (first use of the two loads: here, not in the prologue)
  if constexpr (!TO_SLAB) {   // epilogue operands -> LDS (their region is outside the buffers; published by the barriers below)
    if (tid < 64 * NG) epi_w[tid] = epi_lo | (epi_hi << 16);
    else if (tid - 64 * NG < MT) epi_a[tid - 64 * NG] = epi_lo | (epi_hi << 16);
  }
  __syncthreads();            // the activation buffers are free
  v4i* const red = reinterpret_cast<v4i*>(smem);
  {
    v4i* const mine = red + (size_t)wave * HB * 4 * 64 + lane;
    if (s < 2) {
#pragma unroll
      for (int j = 0; j < HB; ++j)
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) mine[(j * 4 + ab) * 64] = acc[HB + j][ab];
    } else {
#pragma unroll
      for (int j = 0; j < HB; ++j)
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) mine[(j * 4 + ab) * 64] = acc[j][ab];
    }
    __syncthreads();
    const v4i* const theirs = red + (size_t)(wave ^ 2) * HB * 4 * 64 + lane;
    if (s < 2) {
#pragma unroll
      for (int j = 0; j < HB; ++j) {      // (two row blocks' reads in flight: all sixteen next to 128 accumulators spill)
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) acc[j][ab] += theirs[(j * 4 + ab) * 64];
        if (j & 1) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < HB; ++j) {
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) acc[HB + j][ab] += theirs[(j * 4 + ab) * 64];
        if (j & 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
  {
    v4i* const mine = red + (size_t)wave * QB * 4 * 64 + lane;
    const v4i* const theirs = red + (size_t)(wave ^ 1) * QB * 4 * 64 + lane;
    auto park2 = [&](auto send_tag) {
      constexpr int SEND = decltype(send_tag)::value;        // first row block of the quarter handed to the partner
#pragma unroll
      for (int j = 0; j < QB; ++j)
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) mine[(j * 4 + ab) * 64] = acc[SEND + j][ab];
    };
    auto take2 = [&](auto keep_tag) {
      constexpr int KEEP = decltype(keep_tag)::value;        // first row block of the quarter this wave finishes
#pragma unroll
      for (int j = 0; j < QB; ++j) {
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) acc[KEEP + j][ab] += theirs[(j * 4 + ab) * 64];
      }
    };
    if (s == 0) park2(IntTag<QB>{});
    else if (s == 1) park2(IntTag<0>{});
    else if (s == 2) park2(IntTag<HB + QB>{});
    else park2(IntTag<HB>{});
    __syncthreads();
    if (s == 0) take2(IntTag<0>{});
    else if (s == 1) take2(IntTag<QB>{});
    else if (s == 2) take2(IntTag<HB>{});
    else take2(IntTag<HB + QB>{});
  }

  MIDM_STAMP(101);
  // ---- write back: row blocks s * QB .. + QB - 1 of group g -----------------------------------------------------
  // D layout (16x16): col = lane & 15 -> row m of the block, row = (lane >> 4) * 4 + r -> channel slot i.
  // W4: channel = ng * 64 + (i >> 3) * 32 + ab * 8 + (i & 7) (4 consecutive channels per lane); W8: ng * 64 + ab * 16 + i.
  const int mcol = lane & 15;
  const int i0 = (lane >> 4) * 4;
  const bool out16 = !TO_SLAB && p.tile_linear != 0;      // (host: output rows start 16-B aligned)
  auto finish = [&](auto first_tag) {
    constexpr int FIRST = decltype(first_tag)::value;
#pragma unroll
    for (int j = 0; j < QB; ++j) {
      const int mb = FIRST + j;
      const int m = m0 + mb * 16 + mcol;
      float sa = 0.f, as = 0.f;
      if constexpr (!TO_SLAB) {
        const uint32_t av = epi_a[mb * 16 + mcol];
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
  if (s == 0) finish(IntTag<0>{});
  else if (s == 1) finish(IntTag<QB>{});
  else if (s == 2) finish(IntTag<HB>{});
  else finish(IntTag<HB + QB>{});
  MIDM_STAMP(102);
}

}  // namespace omni
