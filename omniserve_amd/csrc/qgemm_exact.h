// Prefill GEMM for EXACT shapes: M % 128 == 0, N % 256 == 0, K % 256 == 0 -- what the models' projections are at prefill.
// (included by qgemm_kernel.h; the generic w4a8_gemm_kernel keeps ragged shapes)
//
// Same tile as the generic prefill kernel (128 rows x 256 channels per 256-thread workgroup, a wave = 64 channels x 128
// rows = 32 accumulator tiles, K in 256-k chunks, two workgroups per CU, XCD-aware tile order) and the same arithmetic
// (bit-identical results), but built as ONE straight path -- branch-free prologue, steady K loop, a last chunk that
// prefetches nothing, packed write-back.  With the generic forms in the same kernel the accumulators are merged across the
// alternative paths (register copies, spills) and the last chunk ran through the predicated form.  What changed on top:
//
//  * ADMA: the activation tile goes global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 4 rows x 256 B per wave
//    instruction): no staging registers (32 VGPRs + 16 of hoisted row pointers in the generic kernel), no ds_write pass.
//    A DMA instruction's LDS image is lane-linear, so (a) the bank swizzle is applied to the SOURCE address -- LDS row m
//    keeps its sixteen 16-B pieces at slot p ^ (m & 15), which makes the B-operand ds_read_b128 of 16 rows x one piece
//    conflict-free -- and (b) the B operand must be 16 CONSECUTIVE k of the row.  The packed int4 tile gives a lane the
//    dwords (tile parity, k5) = 0..3 of ITS k6 = lane >> 4, i.e. four runs of 4 k; a 4 x 4 transpose between the register
//    index and the 16-lane row -- v_permlane32_swap + v_permlane16_swap, 8 swaps per 64-k step on the PACKED registers --
//    turns that into k6 = 0..3 of (tile parity, k5) = lane >> 4 = 16 consecutive k.  (The generic kernel instead stages
//    the activations with four ds_write_b32 per 16-B piece into the weights' k order.)  W8A8 rows are natural already.
//  * write-back: the tile's scale vectors are parked in LDS by the prologue (round 2 read them from global memory inside
//    the write-back loop: 32 dependent round trips per wave); the lane pairs (l, l ^ 16) exchange halves with
//    v_permlane16_swap so that every lane stores 16 B (8 consecutive channels): 16 store instructions per wave instead of 64.
#pragma once

namespace omni {

#ifndef OMNI_GEMM_EXACT_DMA
#define OMNI_GEMM_EXACT_DMA 1
#endif

#ifndef OMNI_GEMM_EXACT_PIN
#define OMNI_GEMM_EXACT_PIN 1
#endif

#ifndef OMNI_GEMM_EXACT_SWP
#define OMNI_GEMM_EXACT_SWP 0     // measured in the ISA only: the fence splits the B-read pipeline (read -> lgkmcnt(0) -> 4 MFMAs)
#endif

#ifdef OMNI_DEBUG_CLOCKS
// timeline probe: per workgroup (wave 0) entry / first chunk / after the K loop / after the stores (100 MHz ticks)
static __device__ unsigned long long omni_dbg_tl[5 * 8192];
#endif

// eight LDS-DMA pieces of one wave: piece i covers the 4 rows (i * 4 .. + 3) of the wave's 32, LDS image lane-linear
// from lds_dst + i * 1024.  voff[i & 3] = the lane's offset inside the 16 rows of (sbase0 | sbase1) (row part + swizzled
// piece); both bases wave-uniform.  One statement: M0 (compiler-reserved) is saved once and restored at the end.
__device__ __forceinline__ void lds_dma16_x8(const void* sbase0, const void* sbase1, uint32_t v0, uint32_t v1, uint32_t v2,
                                             uint32_t v3, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_nop 4\n\ts_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %1\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %2\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %2\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %2\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(sbase0), "s"(sbase1), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(lds_dst)
      : "memory", "scc");
}

// TO_SLAB (decode batches of 256 / 384 / 512 rows: few tiles): the workgroup takes K slice blockIdx.y of p.kslice (a whole
// number of chunks) and stores its int32 accumulators to slab[blockIdx.y][M][N] for splitk_epilogue_kernel / a slab consumer.
template <int MODE, bool ADMA, bool TO_SLAB = false>
__global__ __launch_bounds__(256, 2) void w4a8_gemm_exact_kernel(GemmArgs p) {
  constexpr int MB = 8, MT = 128, WAVES = 4, NTHREADS = 256, A_LOADS = 8;
  static_assert(KCHUNK == 256 && STEPS == 4, "tile maps assume 256-k chunks of four 64-k steps");
  constexpr bool W8C = MODE == MODE_W8 && OMNI_GEMM_W8_COALESCED;    // (see w4a8_gemm_kernel)
  // ONE LDS object (a second one makes hipcc drain vmcnt in front of LDS reads beside DMA traffic): the two activation
  // buffers first -- LDS-DMA destinations stay below 64 KiB --, then the epilogue operands and the W8A8 transpose scratch
  constexpr int LDS_A = 2 * MT * KCHUNK, LDS_EW = 64 * WAVES * 4, LDS_EA = MT * 4, LDS_WT = W8C ? WAVES * 1024 : 0;
  __shared__ __attribute__((aligned(1024))) uint8_t smem[LDS_A + LDS_EW + LDS_EA + LDS_WT];
  uint8_t (*lds)[MT * KCHUNK] = reinterpret_cast<uint8_t (*)[MT * KCHUNK]>(smem);
  uint32_t* const epi_w = reinterpret_cast<uint32_t*>(smem + LDS_A);            // {wscale, w_sz} per channel of the tile
  uint32_t* const epi_a = reinterpret_cast<uint32_t*>(smem + LDS_A + LDS_EW);   // {ascale, asum} per row of the tile
  uint8_t* const wtr = smem + LDS_A + LDS_EW + LDS_EA;
  // one batch of scalar loads for the prologue's kernel arguments (hipcc otherwise fetches them in three dependent groups --
  // tile order, K slice, pointers -- in front of the first tile request)
  asm volatile("" ::"s"(p.A), "s"(p.W), "s"(p.wscales), "s"(p.ascales), "s"(p.wsz), "s"(p.asum), "s"(p.M), "s"(p.N), "s"(p.K), "s"(p.kslice),
               "s"(p.tiles_m), "s"(p.tiles_n), "s"(p.tile_linear), "s"((int)gridDim.x), "s"((int)gridDim.y));

#ifdef OMNI_DEBUG_CLOCKS
  const unsigned long long tl0 = wall_clock64();
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware tile order (w4a8_gemm_kernel): XCD x takes the x-th contiguous eighth of the tiles as 8 x 8 super-blocks
  int tile_m, tile_n;
  {
    const int wid = blockIdx.x;
    if (p.tile_linear) {
      // few row tiles (decode batches of 129 .. 1023 rows): the super-block order would put every existing tile of a block
      // on ONE XCD (only the first rows of each 8 x 8 block exist: 32 tiles of Llama-3-8B's down_proj at bs = 256 ran on 2 of
      // the 8 XCDs, profiles/r04_b); consecutive workgroups -- consecutive XCDs -- take consecutive tiles instead
      tile_m = wid % p.tiles_m;
      tile_n = wid / p.tiles_m;
      if (tile_n >= p.tiles_n) return;
    } else {
    const int per_xcd = gridDim.x >> 3;
    const int t = (wid & 7) * per_xcd + (wid >> 3);
    const int sbn = (p.tiles_n + 7) >> 3;
    const int sb = t >> 6, in = t & 63;
    tile_m = (sb / sbn) * 8 + (in >> 3);
    tile_n = (sb % sbn) * 8 + (in & 7);
    if (tile_m >= p.tiles_m || tile_n >= p.tiles_n) return;
    }
  }
  const int ng = tile_n * WAVES + wave;  // 64-channel group of this wave
  const int m0 = tile_m * MT;
  const int k0 = TO_SLAB ? (int)blockIdx.y * p.kslice : 0;      // first k of this workgroup's K slice
  const int nchunks = (TO_SLAB ? p.kslice : p.K) / KCHUNK;
  // (Two workgroups share a CU, one wave of each per SIMD, and the arbiter strictly favours the OLDER wave: per-workgroup
  // clocks show the first-dispatched workgroup of every CU running its K loop in 30 us and the second in 46, the last 16
  // alone at 2/3 of the paired rate.  s_setprio flips or time-slices that at will -- and the pair finishes at the same
  // time whatever the split: profiles/r03_d.  No priority code here.)

  // ---- weights: HBM / L2 -> VGPR ring (as w4a8_gemm_kernel) ---------------------------------------------
  const int lx = (lane >> 3) & 1, lc = lane & 7, le = lane >> 4;
  const uint8_t* wbase;
  if constexpr (W8C) wbase = p.W + (size_t)(ng * 64 + (lane >> 2)) * p.K + (lane & 3) * 16 + k0;
  else if constexpr (MODE == MODE_W8) wbase = p.W + (size_t)(ng * 64 + (lane & 15)) * p.K + (lane >> 4) * 16 + k0;
  else wbase = p.W + ((size_t)(2 * ng + lx) * (p.K / 32) + k0 / 32) * 512 + (lc * 4 + le) * 16;
  uint8_t* const wtr_w = wtr + (W8C ? wave * 1024 + ((lane & 3) * 16 + ((lane >> 2) ^ (5 * (lane & 3)))) * 16 : 0);
  const uint8_t* const wtr_r = wtr + (W8C ? wave * 1024 + ((lane & 48) + ((lane & 15) ^ (5 * (lane >> 4)))) * 16 : 0);
  auto load_w = [&](int k, int j) -> uint4 {      // k: relative to the K slice
    const uint8_t* ptr;
    if constexpr (MODE == MODE_W8) ptr = wbase + (size_t)j * 16 * p.K + k;
    else ptr = wbase + (size_t)(k / 32 + j) * 512;
    const v4i v = *reinterpret_cast<const v4i*>(ptr);
    return make_uint4((uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]);
  };
  constexpr int WL = (MODE == MODE_W8) ? 4 : 2;  // weight loads per lane per k-step
  constexpr int WRING = MODE == MODE_CHN ? OMNI_GEMM_RING_CHN : OMNI_GEMM_RING_OTHER;
  uint4 wq[WRING][WL];

  // ---- activations ---------------------------------------------------------------------------------------
  // !ADMA: registers -> "plane" image [k-step][16-B slot][row][16 B] in the weights' k order (w4a8_gemm_kernel).
  // ADMA: LDS row m = 256 B, piece q at slot q ^ (m & 15), written by LDS-DMA.
  uint4 areg[ADMA ? 1 : A_LOADS];
  auto piece = [&](int j, int& m, int& kk) {
    const int id = tid + j * NTHREADS;
    if constexpr (MODE == MODE_W8) {
      m = (id & 7) | ((id >> 7) << 3);
      kk = (id >> 3) & 15;
    } else {
      m = ((id >> 2) & 7) | ((id >> 7) << 3);
      kk = (id & 3) | (((id >> 5) & 3) << 2);
    }
  };
  auto load_a = [&](int chunk) {
    if constexpr (!ADMA) {
#pragma unroll
      for (int j = 0; j < A_LOADS; ++j) {
        int m, kk;
        piece(j, m, kk);
        areg[j] = *reinterpret_cast<const uint4*>(p.A + (size_t)(m0 + m) * p.K + k0 + chunk * KCHUNK + kk * 16);
      }
    }
  };
  auto store_a = [&](int buf) {
    if constexpr (!ADMA) {
#pragma unroll
      for (int j = 0; j < A_LOADS; ++j) {
        int m, kk;
        piece(j, m, kk);
        const int ks = kk >> 2;
        if constexpr (MODE == MODE_W8) {
          *reinterpret_cast<uint4*>(&lds[buf][((ks * 4 + (kk & 3)) * MT + m) * 16]) =
              make_uint4(areg[j].x, areg[j].y, areg[j].z, areg[j].w);
        } else {
          const int tp = (kk >> 1) & 1, d = kk & 1;
          uint8_t* dst = &lds[buf][(ks * 4 * MT + m) * 16 + tp * 8 + d * 4];
          *reinterpret_cast<uint32_t*>(dst + 0 * MT * 16) = areg[j].x;
          *reinterpret_cast<uint32_t*>(dst + 1 * MT * 16) = areg[j].y;
          *reinterpret_cast<uint32_t*>(dst + 2 * MT * 16) = areg[j].z;
          *reinterpret_cast<uint32_t*>(dst + 3 * MT * 16) = areg[j].w;
        }
      }
    }
  };
  // DMA lane map: lane -> (row dr = lane >> 4 of the piece's 4 rows, LDS slot ds = lane & 15).  Piece i of wave w covers
  // rows w * 32 + i * 4 + dr; (row & 15) = (i & 3) * 4 + dr, so the source piece is ds ^ ((i & 3) * 4 + dr).
  uint32_t dvo[4] = {0, 0, 0, 0};
  if constexpr (ADMA) {
    const int dr = lane >> 4, ds = lane & 15;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int r = ii * 4 + dr;
      dvo[ii] = (uint32_t)r * (uint32_t)p.K + (uint32_t)((ds ^ r) << 4);
    }
  }
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)smem;
  auto dma_chunk = [&](int chunk, int buf) {
    if constexpr (ADMA) {
      const uint8_t* s0 = reinterpret_cast<const uint8_t*>(p.A) + (size_t)(m0 + wave * 32) * p.K + (size_t)k0 + (size_t)chunk * KCHUNK;
      const uint8_t* s1 = s0 + (size_t)16 * p.K;
      lds_dma16_x8(s0, s1, dvo[0], dvo[1], dvo[2], dvo[3],
                   __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)buf * (MT * KCHUNK) + (uint32_t)wave * 32 * 256));
    }
  };
  // B-operand address of row block 0 at k-step 0 (per lane); row block mb adds an immediate
  uint32_t boff;
  if constexpr (ADMA) boff = (uint32_t)(lane & 15) * 256 + (uint32_t)(((lane >> 4) ^ (lane & 15)) << 4);
  else boff = (uint32_t)((lane >> 4) * MT + (lane & 15)) * 16;

  v4i acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) acc[mb][ab] = (v4i){0, 0, 0, 0};

  // ---- prologue ------------------------------------------------------------------------------------------
  dma_chunk(0, 0);
#pragma unroll
  for (int s = 0; s < WRING; ++s)
#pragma unroll
    for (int j = 0; j < WL; ++j) wq[s][j] = load_w(s * KSTEP, j);
  load_a(0);
  if constexpr (!TO_SLAB) {   // epilogue operands -> LDS (published by the first chunk's barrier); the slab form has none
    const int n = tile_n * 64 * WAVES + tid;
    const uint32_t sw = __builtin_bit_cast(uint16_t, p.wscales[n]);
    uint32_t sz = 0;
    if constexpr (MODE == MODE_CHN) sz = __builtin_bit_cast(uint16_t, p.wsz[n]);
    epi_w[tid] = sw | (sz << 16);
    if (tid < MT) {
      const uint32_t sa = __builtin_bit_cast(uint16_t, p.ascales[m0 + tid]);
      uint32_t as = 0;
      if constexpr (MODE == MODE_CHN) as = __builtin_bit_cast(uint16_t, p.asum[m0 + tid]);
      epi_a[tid] = sa | (as << 16);
    }
  }
  store_a(0);
  if constexpr (ADMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of chunk 0 have landed

  const size_t gcol = (size_t)(2 * ng + lx) * 32 + lc * 4;
  uint32_t gs[2] = {0, 0}, gz[2] = {0, 0};   // per-group second-level params of the current chunk (2 groups of 128)
  if constexpr (MODE == MODE_GRP) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      gs[h] = *reinterpret_cast<const uint32_t*>(p.s2s + (size_t)(k0 / 128 + h) * p.N + gcol);
      gz[h] = *reinterpret_cast<const uint32_t*>(p.s2z + (size_t)(k0 / 128 + h) * p.N + gcol);
    }
  }
#ifdef OMNI_DEBUG_CLOCKS
  const unsigned long long tl1 = wall_clock64();
#endif

  // ---- one K chunk; NEXT = another chunk follows (prefetch its activation tile and this step's weights of it) ----
  // loads issued between this chunk's DMA and the next chunk's barrier that may still be in flight there
  // (the DMA of the next tile goes out at the step 0 / step 1 seam, see below: steps 1..3 refill after it)
  constexpr int VM_AFTER_DMA = (STEPS - 1) * WL;
  auto run_chunk = [&](int c, auto next_tag) {
    constexpr bool NEXT = decltype(next_tag)::value;
    const int kc = c * KCHUNK;
    if constexpr (NEXT) load_a(c + 1);
    uint32_t gsn[2] = {0, 0}, gzn[2] = {0, 0};
    if constexpr (MODE == MODE_GRP && NEXT) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        gsn[h] = *reinterpret_cast<const uint32_t*>(p.s2s + (size_t)((k0 + kc + KCHUNK) / 128 + h) * p.N + gcol);
        gzn[h] = *reinterpret_cast<const uint32_t*>(p.s2z + (size_t)((k0 + kc + KCHUNK) / 128 + h) * p.N + gcol);
      }
    }
    if constexpr (ADMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_AFTER_DMA) : "memory");   // my pieces of chunk c landed
    __syncthreads();          // chunk c is visible in lds[c & 1]; everybody is done reading lds[(c + 1) & 1]
    const uint8_t* abuf = lds[c & 1];

    // unpack (+ lane transpose / per-group dequant) of k-step s into MFMA A operands
    auto unpack = [&](int s, v4i (&wa)[4]) {
      if constexpr (W8C) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {      // LDS is in order within a wave: no wait between the write and the read
          *reinterpret_cast<uint4*>(wtr_w) = make_uint4(wq[s % WRING][rb].x, wq[s % WRING][rb].y, wq[s % WRING][rb].z,
                                                        wq[s % WRING][rb].w);
          wa[rb] = *reinterpret_cast<const v4i*>(wtr_r);
        }
      } else if constexpr (MODE == MODE_W8) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
          wa[rb] = (v4i){(int)wq[s % WRING][rb].x, (int)wq[s % WRING][rb].y, (int)wq[s % WRING][rb].z,
                         (int)wq[s % WRING][rb].w};
      } else {
        // dwords of a 16-B piece: x = (k5 = 0, n2 = 0) y = (0, 1) z = (1, 0) w = (1, 1); d[n2][(tile parity, k5)]
        const uint4 t0 = wq[s % WRING][0], t1 = wq[s % WRING][1];
        uint32_t d[2][4] = {{t0.x, t0.z, t1.x, t1.z}, {t0.y, t0.w, t1.y, t1.w}};
        if constexpr (ADMA) {
          // register index (tile parity, k5) <-> 16-lane row k6: afterwards d[b][q] = k6 = q of (parity, k5) = lane >> 4
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const auto s02 = __builtin_amdgcn_permlane32_swap(d[b][0], d[b][2], false, false);
            const auto s13 = __builtin_amdgcn_permlane32_swap(d[b][1], d[b][3], false, false);
            const auto s01 = __builtin_amdgcn_permlane16_swap((uint32_t)s02[0], (uint32_t)s13[0], false, false);
            const auto s23 = __builtin_amdgcn_permlane16_swap((uint32_t)s02[1], (uint32_t)s13[1], false, false);
            d[b][0] = (uint32_t)s01[0]; d[b][1] = (uint32_t)s01[1]; d[b][2] = (uint32_t)s23[0]; d[b][3] = (uint32_t)s23[1];
          }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            uint32_t u[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) u[q] = (d[b][q] >> (4 * a)) & 0x0F0F0F0Fu;
            if constexpr (MODE == MODE_GRP) {
              const int h = s >> 1;  // group inside the chunk
              const uint32_t sc = (gs[h] >> (8 * (a * 2 + b))) & 0xFFu;
#pragma unroll
              for (int q = 0; q < 4; ++q) u[q] = u[q] * sc;
              vadd4_zbyte_x4(u, gz[h], a * 2 + b);
            }
            wa[a * 2 + b] = (v4i){(int)u[0], (int)u[1], (int)u[2], (int)u[3]};
          }
      }
    };
    // refill step s's weight registers with the step WRING ahead
    auto refill = [&](int s) {
      if (NEXT || s + WRING < STEPS) {
#pragma unroll
        for (int j = 0; j < WL; ++j) wq[s % WRING][j] = load_w(kc + (s + WRING) * KSTEP, j);
        if constexpr (NEXT && !ADMA) __builtin_amdgcn_sched_barrier(0x78F);
      }
    };
    // Explicit one-step software pipeline (SWP): step s + 1 is unpacked inside step s, behind a VALU fence at the step
    // seam (sched_barrier that lets everything but VALU cross).  Left alone hipcc unpacks ALL four steps in front of step 0:
    // 64 operand registers live (the per-group kernel spilled on them) and a VALU burst in front of the first MFMAs.
    constexpr bool SWP = OMNI_GEMM_EXACT_SWP != 0;
    v4i wa[SWP ? 2 : 1][4];
    if constexpr (SWP) { unpack(0, wa[0]); refill(0); }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      // The next tile's DMA goes out HERE, not at the chunk's top: hipcc's counted waits for the weight ring do not see
      // the eight DMA operations, so every vmcnt(N) it emits behind them also drains N-relative DMA pieces.  Issued at the
      // top, the DMA was waited for ~one MFMA later.  The statement orders LDS reads around it, but it lands behind step
      // 0's last B read, three MFMA groups before the seam.
      if constexpr (NEXT) {
        if (s == 1) dma_chunk(c + 1, (c + 1) & 1);
      }
      if constexpr (SWP) {
        __builtin_amdgcn_sched_barrier(0x3FC);
        if (s + 1 < STEPS) { unpack(s + 1, wa[(s + 1) & 1]); refill(s + 1); }
      } else {
        unpack(s, wa[0]);
        refill(s);
      }
      v4i bf[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        if constexpr (ADMA) bf[mb] = *reinterpret_cast<const v4i*>(abuf + (boff ^ (uint32_t)(s << 6)) + mb * 16 * 256);
        else bf[mb] = *reinterpret_cast<const v4i*>(abuf + boff + (s * 4 * MT + mb * 16) * 16);
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int ab = 0; ab < 4; ++ab)
          acc[mb][ab] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[SWP ? (s & 1) : 0][ab], bf[mb], acc[mb][ab], 0, 0, 0);
    }
    if constexpr (OMNI_GEMM_PIPE_B && MODE == MODE_CHN) {   // B-operand reads three row blocks ahead over the whole chunk
      constexpr int PRE = 3;
      __builtin_amdgcn_sched_group_barrier(0x100, PRE, 0);
#pragma unroll
      for (int i = 0; i < STEPS * MB - PRE; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        // the step's weight refill (for the same step of the NEXT chunk) goes out behind the first MFMA group of the step, a
        // chunk of MFMAs ahead of its use: left alone the scheduler sinks all eight loads to the end of the chunk, ~200
        // cycles in front of the wait that needs them (ADMA only: the register-staged form has no registers for it)
        if (NEXT && ADMA && OMNI_GEMM_EXACT_PIN && (i % MB) == 0) __builtin_amdgcn_sched_group_barrier(0x020, WL, 0);
        // (giving the unpack VALU slots in this pipeline -- sched_group_barrier(0x002, 4) per MFMA group -- makes the
        // solver bunch the B reads of a step behind lgkmcnt(0) waits: profiles/r03_d; the compiler's own placement stays)
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * PRE, 0);
    }
    if constexpr (NEXT) store_a((c + 1) & 1);
    if constexpr (MODE == MODE_GRP) {
#pragma unroll
      for (int h = 0; h < 2; ++h) { gs[h] = gsn[h]; gz[h] = gzn[h]; }
    }
  };
  int c = 0;
  for (; c + 1 < nchunks; ++c) run_chunk(c, BoolTag<true>{});
  run_chunk(c, BoolTag<false>{});
  __builtin_amdgcn_sched_barrier(0);     // the write-back (LDS reads, conversions) stays behind the last MFMAs
#ifdef OMNI_DEBUG_CLOCKS
  const unsigned long long tl2 = wall_clock64();
#endif

  // ---- write back ----------------------------------------------------------------------------------------
  // D layout (16x16): col = lane & 15 -> row m of the block, row = (lane >> 4) * 4 + r -> channel slot i.
  // W4: i = x * 8 + c, channel = ng * 64 + x * 32 + ab * 8 + c (4 consecutive channels per lane); W8: ng * 64 + ab * 16 + i.
  const int mcol = lane & 15;
  const int i0 = (lane >> 4) * 4;
  if constexpr (TO_SLAB) {      // int32 accumulators of this K slice -> slab[blockIdx.y] (4 consecutive channels per lane: 16-B stores)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      int32_t* row = p.slab + ((size_t)blockIdx.y * p.M + (m0 + mb * 16 + mcol)) * p.N;
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) {
        int n;
        if constexpr (MODE == MODE_W8) n = ng * 64 + ab * 16 + i0;
        else n = ng * 64 + (i0 >> 3) * 32 + ab * 8 + (i0 & 7);
        *reinterpret_cast<v4i*>(row + n) = acc[mb][ab];
      }
    }
    return;
  }
  // The lane's 16 channels are the same for every row block: their scales are converted once (32 registers the K loop
  // no longer needs); the arithmetic runs on float pairs (v_pk_mul_f32 / v_pk_add_f32: the reference's three products and one
  // difference, each rounded to f32 -- -ffp-contract=off) and the pair is made opaque before the fp16 conversion: hipcc
  // otherwise folds `(half)(f32 * f32)` into v_fma_mixlo_f16 -- ONE rounding to fp16 where the reference rounds to f32 first --
  // and a handful of outputs per million differ by one fp16 ulp (caught by the cross-build checksum of tools/gemm_ab.py).
  // ~3.5 VALU per output instead of ~10.
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef _Float16 v2h __attribute__((ext_vector_type(2)));
  v2f swf[4][2], szf[4][2];
#pragma unroll
  for (int ab = 0; ab < 4; ++ab) {
    int nl;
    if constexpr (MODE == MODE_W8) nl = wave * 64 + ab * 16 + i0;
    else nl = wave * 64 + (i0 >> 3) * 32 + ab * 8 + (i0 & 7);
    const uint4 w4 = *reinterpret_cast<const uint4*>(&epi_w[nl]);      // {wscale, w_sz} x 4 channels
    const uint32_t wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      swf[ab][h] = (v2f){(float)__builtin_bit_cast(half_t, (uint16_t)(wv[2 * h] & 0xFFFFu)),
                         (float)__builtin_bit_cast(half_t, (uint16_t)(wv[2 * h + 1] & 0xFFFFu))};
      szf[ab][h] = (v2f){(float)__builtin_bit_cast(half_t, (uint16_t)(wv[2 * h] >> 16)),
                         (float)__builtin_bit_cast(half_t, (uint16_t)(wv[2 * h + 1] >> 16))};
    }
  }
  auto finish4 = [&](const v4i a4, int ab, float sa, float as) -> uint2 {     // 4 consecutive channels of one row -> 4 fp16
    uint32_t o[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const v2f f = (v2f){(float)a4[2 * h], (float)a4[2 * h + 1]};
      v2f r;
      if constexpr (MODE == MODE_CHN) {
        v2f t = f * swf[ab][h];
        t = t * sa;
        const v2f c = szf[ab][h] * as;
        r = t - c;
      } else {
        const v2f sc = swf[ab][h] * sa;
        r = f * sc;
      }
      asm volatile("" : "+v"(r));
      o[h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, v2h));
    }
    return make_uint2(o[0], o[1]);
  };
  const int odd = (lane >> 4) & 1;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = m0 + mb * 16 + mcol;
    const uint32_t av = epi_a[mb * 16 + mcol];                         // {ascale, asum} of the row
    const float sa = (float)__builtin_bit_cast(half_t, (uint16_t)(av & 0xFFFFu));
    const float as = (float)__builtin_bit_cast(half_t, (uint16_t)(av >> 16));
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const uint2 x = finish4(acc[mb][2 * pr], 2 * pr, sa, as), y = finish4(acc[mb][2 * pr + 1], 2 * pr + 1, sa, as);
      const auto lo = __builtin_amdgcn_permlane16_swap(x.x, y.x, false, false);
      const auto hi = __builtin_amdgcn_permlane16_swap(x.y, y.y, false, false);
      int n8;
      if constexpr (MODE == MODE_W8) n8 = ng * 64 + (2 * pr + odd) * 16 + (lane >> 5) * 8;
      else n8 = ng * 64 + (lane >> 5) * 32 + (2 * pr + odd) * 8;
      *reinterpret_cast<uint4*>(p.out + (size_t)m * p.out_stride + n8) =
          make_uint4((uint32_t)lo[0], (uint32_t)hi[0], (uint32_t)lo[1], (uint32_t)hi[1]);
    }
  }
#ifdef OMNI_DEBUG_CLOCKS
  if (tid == 0 && blockIdx.x < 8192) {
    uint32_t hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    omni_dbg_tl[blockIdx.x * 5 + 0] = tl0;
    omni_dbg_tl[blockIdx.x * 5 + 1] = tl1;
    omni_dbg_tl[blockIdx.x * 5 + 2] = tl2;
    omni_dbg_tl[blockIdx.x * 5 + 3] = wall_clock64();
    omni_dbg_tl[blockIdx.x * 5 + 4] = ((unsigned long long)xcc_id() << 32) | hwid;   // (wave 0's: slot bits 3:0, SIMD 5:4, CU 11:8)
  }
#endif
}

}  // namespace omni
