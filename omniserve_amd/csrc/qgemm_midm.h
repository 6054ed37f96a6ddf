// Mid-M kernel (M = 33 .. 128 rows per tile; grid.z row tiles beyond): the decode GEMMs at the batches between the
// single-wave GEMV tiles (M <= 32) and the MFMA-bound prefill tile (M > 256).  Included by qgemm_kernel.h.
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
//     s of every 256-k chunk for group g and all rows.  A weight byte is loaded once, by one wave, straight into VGPRs
//     (the packed tile IS the MFMA A operand, as everywhere in this library; 1-KiB coalesced non-temporal wave loads, a
//     ring of R chunks per wave = 8 x R x 2 KiB in flight per CU); the unpack runs once per byte (40 VALU per 32 MFMAs
//     at 128 rows -- the 32-row tiles spend 40 per 8);
//   * the activation chunk (rows x 256 B) goes global -> LDS by LDS-DMA, once per workgroup, double buffered, one barrier
//     per chunk; all 8 waves read their k-step of it as MFMA B operands (image and lane transposition of the packed
//     registers as in w4a8_gemm_exact_kernel: LDS row m keeps piece q at slot q ^ (m & 15));
//   * the four K-phase partials of a group meet in LDS after the K loop (two exchange rounds over the activation
//     buffers' memory, static accumulator indices only); wave (g, s) finishes row quarter s: epilogue to fp16, or the
//     int32 slab of its K slice for the slab consumers / splitk_epilogue_kernel.
// Activation traffic per launch = (N / 128) x M x K bytes from L2 (2x the weight bytes at M = 128), weights 1x from HBM.
#pragma once

namespace omni {

#ifndef OMNI_MIDM_RING
// chunks of weights in flight per wave.  Loads return in order and every chunk's top waits for this wave's DMA pieces,
// i.e. for every refill but the newest: a deeper ring holds nothing more in flight (and 3 .. 4 slots spilled at 128 rows).
#define OMNI_MIDM_RING 2
#endif
#ifndef OMNI_MIDM_PIPE
#define OMNI_MIDM_PIPE 1          // B-operand reads pinned PRE row blocks ahead of their MFMAs (sched_group_barrier)
#endif

// NI LDS-DMA pieces of one wave (piece = 4 rows x 256 B, LDS image lane-linear from lds_dst + i * 1024), one scalar base
// (the chunk's first byte of row 0) + a per-piece lane offset (row * K + swizzled 16-B piece).  One statement: M0
// (compiler-reserved) is saved once and restored at the end.
__device__ __forceinline__ void lds_dma16_x2(const void* sbase, uint32_t v0, uint32_t v1, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_nop 4\n\ts_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(sbase), "v"(v0), "v"(v1), "s"(lds_dst)
      : "memory", "scc");
}
__device__ __forceinline__ void lds_dma16_x4(const void* sbase, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3,
                                             uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_nop 4\n\ts_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(sbase), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(lds_dst)
      : "memory", "scc");
}

template <int V> struct IntTag { static constexpr int value = V; };

template <int MB, int MODE, bool TO_SLAB, bool NT>
__global__ __launch_bounds__(512, 2) void w4a8_midm_kernel(GemmArgs p) {
  static_assert(MB == 4 || MB == 8, "row tile of 64 or 128 rows (quarters of whole 16-row blocks)");
  constexpr int MT = MB * 16;
  constexpr int NG = 2, NW = 8;
  constexpr int CH = KCHUNK;                               // k per chunk: four 64-k steps, one per K phase
  constexpr int NI = MT / 32;                              // DMA pieces per wave and chunk (MT rows / 8 waves / 4 rows)
  constexpr int WL = (MODE == MODE_W8) ? 4 : 2;
  constexpr int GP = (MODE == MODE_GRP) ? 2 : 0;           // second-level parameter loads per k-step
  constexpr int R = OMNI_MIDM_RING;
  constexpr int HB = MB / 2, QB = MB / 4;
  constexpr int LDS_A = 2 * MT * CH;                       // two activation buffers (LDS-DMA destinations stay below 64 KiB)
  constexpr int LDS_RED = NW * HB * 4 * 1024;              // first exchange round: every wave parks half of its accumulators
  constexpr int LDS_MAIN = LDS_RED > LDS_A ? LDS_RED : LDS_A;
  constexpr int LDS_EPI = TO_SLAB ? 0 : (64 * NG + MT) * 4;
  static_assert(STEPS == 4 && KCHUNK == 256, "one k-step of a 256-k chunk per K phase");
  __shared__ __attribute__((aligned(1024))) uint8_t smem[LDS_MAIN + LDS_EPI + 16];
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

  // ---- weights: HBM -> VGPR ring, one k-step per chunk ----------------------------------------------------
  const int lx = (lane >> 3) & 1, lc = lane & 7, le = lane >> 4;
  const uint8_t* wbase;
  if constexpr (MODE == MODE_W8) wbase = p.W + (size_t)(ng * 64 + (lane & 15)) * p.K + (lane >> 4) * 16 + k0 + s * KSTEP;
  else wbase = p.W + ((size_t)(2 * ng + lx) * (p.K / 32) + (k0 + s * KSTEP) / 32) * 512 + (lc * 4 + le) * 16;
  auto load_w = [&](int c, int j) -> uint4 {      // chunk c of the slice; W4: j = tile parity, W8: j = 16-row block
    const uint8_t* ptr;
    if constexpr (MODE == MODE_W8) ptr = wbase + (size_t)j * 16 * p.K + (size_t)c * CH;
    else ptr = wbase + (size_t)(c * (CH / 32) + j) * 512;
    v4i v;
    if constexpr (NT) v = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(ptr));
    else v = *reinterpret_cast<const v4i*>(ptr);
    return make_uint4((uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]);
  };
  const size_t gcol = (size_t)(2 * ng + lx) * 32 + lc * 4;
  auto load_gp = [&](const uint8_t* base, int c) -> uint32_t {     // 128-k group of (chunk c, phase s)
    return *reinterpret_cast<const uint32_t*>(base + (size_t)(k0 / 128 + 2 * c + (s >> 1)) * p.N + gcol);
  };
  uint4 wq[R][WL];
  uint32_t gs[R], gz[R];

  // ---- activations: LDS-DMA, LDS row m = 256 B with piece q at slot q ^ (m & 15) -----------------------------
  uint32_t dvo[NI];
  {
    const int dr = lane >> 4, ds = lane & 15;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int rl = wave * (MT / NW) + i * 4 + dr;                 // row inside the tile
      const int row = (m0 + rl) < p.M ? (m0 + rl) : (p.M - 1);      // rows beyond M re-read the last row (never stored)
      dvo[i] = (uint32_t)row * (uint32_t)p.K + (uint32_t)((ds ^ (rl & 15)) << 4);
    }
  }
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)smem;
  auto dma_chunk = [&](int c) {
    const uint8_t* sb = reinterpret_cast<const uint8_t*>(p.A) + (size_t)k0 + (size_t)c * CH;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(c & 1) * (MT * CH) + (uint32_t)wave * (MT / NW) * 256);
    if constexpr (NI == 4) lds_dma16_x4(sb, dvo[0], dvo[1], dvo[2], dvo[3], dst);
    else lds_dma16_x2(sb, dvo[0], dvo[1], dst);
  };
  // B operand of row block 0 at this wave's k-step (per lane); row block mb adds an immediate
  const uint32_t boff = (uint32_t)(lane & 15) * 256 + ((uint32_t)(((lane >> 4) ^ (lane & 15)) << 4) ^ (uint32_t)(s << 6));

  v4i acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) acc[mb][ab] = (v4i){0, 0, 0, 0};

  // ---- prologue ------------------------------------------------------------------------------------------
  dma_chunk(0);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int cr = r < nchunks ? r : nchunks - 1;       // (short slices) re-read the last chunk, never consumed
#pragma unroll
    for (int j = 0; j < WL; ++j) wq[r][j] = load_w(cr, j);
    if constexpr (MODE == MODE_GRP) { gs[r] = load_gp(p.s2s, cr); gz[r] = load_gp(p.s2z, cr); }
    else { gs[r] = 0; gz[r] = 0; }
  }
  if constexpr (!TO_SLAB) {   // epilogue operands -> LDS (published by the first chunk's barrier)
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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // chunk 0's pieces (and the head of the ring) have landed

  // unpack (+ lane transposition / per-group dequant) of one k-step into the four MFMA A operands
  auto unpack = [&](const uint4 (&w)[WL], uint32_t sc4, uint32_t zr4, v4i (&wa)[4]) {
    if constexpr (MODE == MODE_W8) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) wa[rb] = (v4i){(int)w[rb].x, (int)w[rb].y, (int)w[rb].z, (int)w[rb].w};
    } else {
      // dwords of a 16-B piece: x = (k5 = 0, n2 = 0) y = (0, 1) z = (1, 0) w = (1, 1); d[n2][(tile parity, k5)]
      const uint4 t0 = w[0], t1 = w[1];
      uint32_t d[2][4] = {{t0.x, t0.z, t1.x, t1.z}, {t0.y, t0.w, t1.y, t1.w}};
      // (explicit copies out of the ring registers: the swaps below work in place, and with the ring slot itself as their
      //  operand the allocator carries the slot across the loop's back edge with two components exchanged -- the refill is
      //  then loaded elsewhere and copied in behind a vmcnt(0).  Eight moves per 32 MFMAs.)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("v_mov_b32 %0, %1" : "=v"(d[b][q]) : "v"(d[b][q]));
      // register index (tile parity, k5) <-> 16-lane row k6: afterwards d[b][q] = k6 = q of (parity, k5) = lane >> 4,
      // i.e. 16 consecutive k per lane -- what a DMA-written activation row offers (w4a8_gemm_exact_kernel)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const auto s02 = __builtin_amdgcn_permlane32_swap(d[b][0], d[b][2], false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(d[b][1], d[b][3], false, false);
        const auto s01 = __builtin_amdgcn_permlane16_swap((uint32_t)s02[0], (uint32_t)s13[0], false, false);
        const auto s23 = __builtin_amdgcn_permlane16_swap((uint32_t)s02[1], (uint32_t)s13[1], false, false);
        d[b][0] = (uint32_t)s01[0]; d[b][1] = (uint32_t)s01[1]; d[b][2] = (uint32_t)s23[0]; d[b][3] = (uint32_t)s23[1];
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          uint32_t u[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) u[q] = (d[b][q] >> (4 * a)) & 0x0F0F0F0Fu;
          if constexpr (MODE == MODE_GRP) {
            const uint32_t sc = (sc4 >> (8 * (a * 2 + b))) & 0xFFu;
#pragma unroll
            for (int q = 0; q < 4; ++q) u[q] = u[q] * sc;
            vadd4_zbyte_x4(u, zr4, a * 2 + b);
          }
          wa[a * 2 + b] = (v4i){(int)u[0], (int)u[1], (int)u[2], (int)u[3]};
        }
    }
  };

  // ---- one chunk: this wave's k-step of it.  RS = ring slot (static); STEADY = a next chunk exists and the refill is
  // issued unconditionally (clamped to the slice): no control flow, every compiler wait is a counted vmcnt ---------
  constexpr int VM_AFTER_DMA = WL + GP;       // loads issued behind a chunk's DMA before the next chunk's top (steady)
  auto body = [&](int c, auto slot_tag, auto steady_tag) {
    constexpr int RS = decltype(slot_tag)::value;
    constexpr bool STEADY = decltype(steady_tag)::value;
    // my pieces of chunk c have landed (steady: behind them only the previous body's refill; tail: nothing)
    if constexpr (STEADY) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_AFTER_DMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();          // chunk c is visible in buffer c & 1; everybody is done reading buffer (c + 1) & 1
    const uint8_t* abuf = smem + (c & 1) * (MT * CH);
    v4i wa[4];
    unpack(wq[RS], gs[RS], gz[RS], wa);
    // the next tile's DMA goes out BEHIND the unpack: hipcc's counted waits for the ring slot do not see DMA operations, so
    // a vmcnt(N) behind fresh pieces also waits for N-relative pieces (w4a8_gemm_exact_kernel: same finding)
    if (STEADY || c + 1 < nchunks) dma_chunk(c + 1);
    if constexpr (STEADY) {
      const int cn = c + R < nchunks ? c + R : nchunks - 1;
      // (fence in front as well: with a refill scheduled among the unpack's reads of the slot it replaces, the slot gets a
      //  second register tuple and the loaded value is COPIED into the loop-carried one -- a vmcnt(0) right behind the load)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < WL; ++j) wq[RS][j] = load_w(cn, j);
      if constexpr (MODE == MODE_GRP) { gs[RS] = load_gp(p.s2s, cn); gz[RS] = load_gp(p.s2z, cn); }
      // the refill stays HERE, R chunks of MFMAs ahead of its use (left alone it sinks to the end of the body, and the
      // next body's unpack rises above the loop's back edge: register copies of in-flight loads, vmcnt(0) per round)
      __builtin_amdgcn_sched_barrier(0);
    }
    v4i bf[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) bf[mb] = *reinterpret_cast<const v4i*>(abuf + boff + mb * 16 * 256);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int ab = 0; ab < 4; ++ab)
        acc[mb][ab] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[ab], bf[mb], acc[mb][ab], 0, 0, 0);
#if OMNI_MIDM_PIPE
    {   // B reads PRE row blocks ahead of the MFMAs that use them (left alone: read, read, wait, 8 MFMAs, ...)
      constexpr int PRE = MB < 3 ? MB : 3;
      __builtin_amdgcn_sched_group_barrier(0x100, PRE, 0);
#pragma unroll
      for (int i = 0; i < MB - PRE; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * PRE, 0);
    }
#endif
  };
  {
    int c = 0;
    for (; c + R < nchunks; c += R) {      // whole ring rounds with a chunk behind them
      body(c, IntTag<0>{}, BoolTag<true>{});
      if constexpr (R > 1) body(c + 1, IntTag<1 % R>{}, BoolTag<true>{});
      if constexpr (R > 2) body(c + 2, IntTag<2 % R>{}, BoolTag<true>{});
      if constexpr (R > 3) body(c + 3, IntTag<3 % R>{}, BoolTag<true>{});
      static_assert(R <= 4, "ring rounds are unrolled by hand");
    }
    const int rem = nchunks - c;            // 1 .. R chunks left: nothing to refill
    body(c, IntTag<0>{}, BoolTag<false>{});
    if (R > 1 && rem > 1) body(c + 1, IntTag<1 % R>{}, BoolTag<false>{});
    if (R > 2 && rem > 2) body(c + 2, IntTag<2 % R>{}, BoolTag<false>{});
    if (R > 3 && rem > 3) body(c + 3, IntTag<3 % R>{}, BoolTag<false>{});
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- the four K phases of a group meet in LDS (static accumulator indices only) ------------------------------
  // round 1: phases {0,1} keep the low half of the row blocks and park the high half, phases {2,3} the other way round;
  //          partner = phase ^ 2.  round 2: inside {0,1} and {2,3} the same with quarters; partner = phase ^ 1.
  // Afterwards wave (g, s) holds the finished accumulators of row blocks s * QB .. s * QB + QB - 1.
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
      for (int j = 0; j < HB; ++j) {      // (one row block at a time: sixteen reads in flight next to 128 accumulators spill)
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) acc[j][ab] += theirs[(j * 4 + ab) * 64];
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < HB; ++j) {
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) acc[HB + j][ab] += theirs[(j * 4 + ab) * 64];
        __builtin_amdgcn_sched_barrier(0);
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
        __builtin_amdgcn_sched_barrier(0);
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

  // ---- write back: row blocks s * QB .. + QB - 1 of group g -----------------------------------------------------
  // D layout (16x16): col = lane & 15 -> row m of the block, row = (lane >> 4) * 4 + r -> channel slot i.
  // W4: channel = ng * 64 + (i >> 3) * 32 + ab * 8 + (i & 7) (4 consecutive channels per lane); W8: ng * 64 + ab * 16 + i.
  const int mcol = lane & 15;
  const int i0 = (lane >> 4) * 4;
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
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) {
        int nl;     // channel inside the workgroup's tile
        if constexpr (MODE == MODE_W8) nl = g * 64 + ab * 16 + i0;
        else nl = g * 64 + (i0 >> 3) * 32 + ab * 8 + (i0 & 7);
        const int n = blockIdx.x * 64 * NG + nl;
        const v4i a4 = acc[mb][ab];
        if (m >= p.M) continue;
        if constexpr (TO_SLAB) {
          int32_t* dst = p.slab + ((size_t)blockIdx.y * p.M + m) * p.N + n;
          *reinterpret_cast<v4i*>(dst) = a4;
        } else {
          const uint4 w4 = *reinterpret_cast<const uint4*>(&epi_w[nl]);      // {wscale, w_sz} x 4 channels
          const uint32_t wv[4] = {w4.x, w4.y, w4.z, w4.w};
          half_t o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            o[r] = epilogue<MODE>(a4[r], (float)__builtin_bit_cast(half_t, (uint16_t)(wv[r] & 0xFFFFu)), sa,
                                  (float)__builtin_bit_cast(half_t, (uint16_t)(wv[r] >> 16)), as);
          *reinterpret_cast<uint2*>(p.out + (size_t)m * p.out_stride + n) = *reinterpret_cast<const uint2*>(o);
        }
      }
    }
  };
  if (s == 0) finish(IntTag<0>{});
  else if (s == 1) finish(IntTag<QB>{});
  else if (s == 2) finish(IntTag<HB>{});
  else finish(IntTag<HB + QB>{});
}

}  // namespace omni
