// W4A8 (per-channel / per-group) and W8A8 GEMM for MI355X (gfx950), hand-written HIP.
//
// Replaces omniserve_backend.qgemm_w4a8_per_chn / qgemm_w4a8_per_group / qgemm_w8a8
// (reference: kernels/csrc/qgemm/*/gemm_cuda.cu).  Not a translation of the CUDA kernels:
//
//  * The packed weight tile (32 n x 32 k codes = 512 B, lane = n3*4+k6, byte = k5*8+n2*4+k7,
//    nibble n1; w4a8_linear.py:296-327) was shaped for mma.m16n8k32 fragments.  Here a wave
//    streams two tile rows (64 output channels) straight from HBM into VGPRs with one
//    16-B load per lane, and the unpacked dwords ARE the A operand of v_mfma_i32_16x16x64_i8:
//    lane L takes chunk (n3 = L&7, k6 = L>>4) of tile row 2*ng + ((L>>3)&1); the four
//    (n1,n2) row sets of a chunk become four MFMAs.  The dot product is invariant under a
//    permutation of k shared by both operands, so the int8 activations are staged into LDS
//    in the matching k order (no cross-lane shuffles, no LDS round trip for weights).
//  * Activations (shared by the workgroup's waves) go HBM/L2 -> VGPR -> LDS once per 256-k
//    chunk, double buffered, one barrier per chunk; weights are prefetched one chunk ahead.
//  * Decode shapes (M <= 128) split K over workgroups so that all 256 CUs stream weights;
//    partial int32 tiles go to a scratch slab and a tiny epilogue kernel reduces them
//    (integer addition: exact and order independent).
//  * Epilogues are compiled with -ffp-contract=off in the reference's operand order so the
//    fp16 results are bit-identical to oracle/w4a8.py.
#pragma once
#include "common.h"
#include "row_reduce.h"

namespace omni {

constexpr int KSTEP = 64;    // k consumed by one round of MFMAs
constexpr int KCHUNK = 256;  // k staged in LDS per barrier
constexpr int STEPS = KCHUNK / KSTEP;

enum { MODE_CHN = 0, MODE_GRP = 1, MODE_W8 = 2 };
template <bool B> struct BoolTag { static constexpr bool value = B; };

struct GemmArgs {
  const int8_t* A;        // [M,K]
  const uint8_t* W;       // packed [N,K/2] (W4) or [N,K] (W8)
  const uint8_t* s2s;     // [K/128, N] per-group scales (unsigned bytes)
  const uint8_t* s2z;     // [K/128, N] per-group zeros
  const half_t* wscales;  // [N]
  const half_t* ascales;  // [M]
  const half_t* wsz;      // [N]
  const half_t* asum;     // [M]
  half_t* out;            // [M, out_stride]
  int32_t* slab;          // [SK][M][N] when split
  int M, N, K;
  int64_t out_stride;
  int kslice;             // k per split (multiple of 64)
  int tiles_m, tiles_n;   // > 0: 1-D grid with the XCD-aware tile order of w4a8_gemm_kernel
  int tile_linear;        // 1 (with tiles_n > 0): workgroup w = tile (w % tiles_m, w / tiles_m) -- few row tiles
  // ---- row-kernel-free decode forms (fused extension; see w4a8_gemv_kernel EPI / A16) ----
  const half_t* A16;      // [M,K] fp16 activations quantised on the fly (A16 kernels; `A` unused)
  uint32_t* amax;         // [AMAX_WORDS] row-maximum candidates (common.h): EPI = 1 raises them, A16 reads them
  half_t* sum_out;        // [M] A16 rider outputs: the reference-ordered fp16 row sum (NULL = not wanted) ...
  half_t* scale_out;      // [M] ... and h(amax / 127)
#ifdef OMNI_TUNING
  int dbg;                // timing experiments (OMNI_GEMV_DBG; wrong results): 1 no rider work, 2 no conversion, 4 no row-maximum atomics
#endif
};
// ablation switches of the fp16-input / SiLU-epilogue GEMVs: tuning builds only (the release kernels carry no such branch)
#ifdef OMNI_TUNING
static inline int gemv_dbg_flags() {
  static const int v = omni_knob("OMNI_GEMV_DBG", 0);
  return v;
}
#define OMNI_GEMV_DBG_BIT(p, bit) (((p).dbg & (bit)) != 0)
#define OMNI_GEMV_SET_DBG(a) ((a).dbg = gemv_dbg_flags())
#else
#define OMNI_GEMV_DBG_BIT(p, bit) false
#define OMNI_GEMV_SET_DBG(a) ((void)0)
#endif


// per-byte add mod 256 (CUDA __vadd4)
__device__ __forceinline__ uint32_t vadd4(uint32_t x, uint32_t y) {
  return ((x & 0x7f7f7f7fu) + (y & 0x7f7f7f7fu)) ^ ((x ^ y) & 0x80808080u);
}
// __vadd4(x, z * 0x01010101) with z = byte ZB of `zw` -- the second-level zero point of one (group, channel), four of which
// sit in one dword of the permuted parameter array: four SDWA byte adds (each keeps the low byte of x.byte_j + z and leaves
// the other bytes alone) instead of extracting z, replicating it with a quarter-rate multiply and the six-operation
// carry-free add.  The per-group kernels are VALU bound (184 op-equivalents per 64-k step against 32 MFMAs).
#ifndef OMNI_GRP_SDWA
#define OMNI_GRP_SDWA 1
#endif
// Four words at a time, byte by byte across the words: on gfx950 a VALU result written with dst_sel != DWORD may not be read
// by the very next VALU instruction (one wait state; hipcc pads its own code, not the inside of an asm statement) -- the first
// version, four byte adds per word back to back, returned wrong bytes.  Interleaved, three instructions separate the writes
// of one register; the statement's outputs are read by compiler code, which gets hipcc's boundary pad.
// The statement is generated: byte j of word q + byte Z of the zero-point dword, q = 0..3 inside j = 0..3 (the first write
// of a register pads the other bytes, the later ones preserve them).
#define OMNI_VADD4_LINE(R, S, J, UNUSED, Z) \
  "v_add_u32_sdwa %" #R ", %" #S ", %8 dst_sel:BYTE_" #J " dst_unused:" UNUSED " src0_sel:BYTE_" #J " src1_sel:BYTE_" #Z "\n\t"
#define OMNI_VADD4_BYTE(J, UNUSED, Z) \
  OMNI_VADD4_LINE(0, 4, J, UNUSED, Z) OMNI_VADD4_LINE(1, 5, J, UNUSED, Z) OMNI_VADD4_LINE(2, 6, J, UNUSED, Z) OMNI_VADD4_LINE(3, 7, J, UNUSED, Z)
#define OMNI_VADD4_ASM(Z) \
  OMNI_VADD4_BYTE(0, "UNUSED_PAD", Z) OMNI_VADD4_BYTE(1, "UNUSED_PRESERVE", Z) OMNI_VADD4_BYTE(2, "UNUSED_PRESERVE", Z) OMNI_VADD4_BYTE(3, "UNUSED_PRESERVE", Z)
#define OMNI_VADD4_STMT(Z) \
  asm(OMNI_VADD4_ASM(Z) : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(zw))
template <int ZB>
__device__ __forceinline__ void vadd4_zbyte_x4(uint32_t (&u)[4], uint32_t zw) {
#if OMNI_GRP_SDWA
  uint32_t r0, r1, r2, r3;
  if constexpr (ZB == 0) OMNI_VADD4_STMT(0);
  else if constexpr (ZB == 1) OMNI_VADD4_STMT(1);
  else if constexpr (ZB == 2) OMNI_VADD4_STMT(2);
  else OMNI_VADD4_STMT(3);
  u[0] = r0; u[1] = r1; u[2] = r2; u[3] = r3;
#else
  const uint32_t zr = ((zw >> (8 * ZB)) & 0xFFu) * 0x01010101u;
#pragma unroll
  for (int q = 0; q < 4; ++q) u[q] = vadd4(u[q], zr);
#endif
}
// (zb is a constant after unrolling: one statement survives)
__device__ __forceinline__ void vadd4_zbyte_x4(uint32_t (&u)[4], uint32_t zw, int zb) {
  switch (zb) {
    case 0: vadd4_zbyte_x4<0>(u, zw); break;
    case 1: vadd4_zbyte_x4<1>(u, zw); break;
    case 2: vadd4_zbyte_x4<2>(u, zw); break;
    default: vadd4_zbyte_x4<3>(u, zw); break;
  }
}

// (the f32 result is made opaque before the fp16 conversion: hipcc otherwise may fold `(half)(f32 * f32)` into
//  v_fma_mixlo_f16 -- ONE rounding where the reference rounds to f32 first; one output in ~10^4 then differs by an ulp,
//  depending on the code around the call: seen in w4a8_gemm_exact_kernel (round 3) and in w4a8_midm_kernel's store forms)
template <int MODE>
__device__ __forceinline__ half_t epilogue(int acc, float sw, float sa, float sz, float asum) {
  if constexpr (MODE == MODE_CHN) {
    float t = (float)acc * sw;
    t = t * sa;
    float c = sz * asum;
    return (half_t)rounded_f32(t - c);
  } else {
    float s = sw * sa;
    return (half_t)rounded_f32((float)acc * s);
  }
}

// One workgroup = WAVES waves; wave w owns output channels [64*(WAVES*bx + w), +64);
// all waves share the LDS-staged activation tile of MB*16 rows.
#ifndef OMNI_GEMM_XCD_ORDER
#define OMNI_GEMM_XCD_ORDER 1
#endif
#ifndef OMNI_GEMM_MIN_BLOCKS
#define OMNI_GEMM_MIN_BLOCKS 2
#endif
#ifndef OMNI_GEMM_RING_OTHER
#define OMNI_GEMM_RING_OTHER 2
#endif
#ifndef OMNI_GEMM_RING_CHN
#define OMNI_GEMM_RING_CHN 4     // a whole chunk ahead; 2 measured 4-6 % slower for the per-channel kernel
#endif
#ifndef OMNI_GEMM_RING_W8
#define OMNI_GEMM_RING_W8 2      // W8A8 on the ragged-shape 128-row tile: k-steps of weights ahead.  2 ends 3 - 7 VGPRs over the budget
                                 // (parked in scratch outside the loops, tests/test_code_objects_cpu.py) and still measures 5 - 7 %
                                 // faster than the spill-free ring of 1 (tools/ragged_gemm_ab.py: 1132 vs 1060 TOPS at M = 4100)
#endif
#ifndef OMNI_GEMV_EPI_PARK
#define OMNI_GEMV_EPI_PARK 1     // 64-row GEMV tiles with the in-kernel epilogue: its operands parked in LDS across the K loop
#endif
#ifndef OMNI_GEMM_PIPE_B
#define OMNI_GEMM_PIPE_B 1
#endif
#ifndef OMNI_GEMM_STORE_A_STEP
#define OMNI_GEMM_STORE_A_STEP 4     // 0..3: publish after that step; 4: at the chunk end (3 measured equal, 1-2 twice slower: the loads are not back yet)
#endif
#ifndef OMNI_GEMM_GRP_STEADY
#define OMNI_GEMM_GRP_STEADY 1
#endif
#ifndef OMNI_GEMM_W8_COALESCED
#define OMNI_GEMM_W8_COALESCED 1     // W8A8 prefill tile: row-coalesced weight loads + lane transpose through LDS (see the kernel)
#endif
template <int MB, int MODE, int WAVES, bool TO_SLAB, bool NT>
__global__ __launch_bounds__(64 * WAVES, OMNI_GEMM_MIN_BLOCKS) void w4a8_gemm_kernel(GemmArgs p) {
  constexpr int MT = MB * 16;
  constexpr int NTHREADS = 64 * WAVES;
  constexpr int A_LOADS = (MT * KCHUNK / 16 + NTHREADS - 1) / NTHREADS;  // 16-B pieces per thread
  __shared__ __attribute__((aligned(16))) uint8_t lds[2][MT * KCHUNK];
  // (one batch of scalar loads for the prologue's kernel arguments, as in w4a8_gemm_exact_kernel)
  asm volatile("" ::"s"(p.A), "s"(p.W), "s"(p.wscales), "s"(p.ascales), "s"(p.wsz), "s"(p.asum), "s"(p.M), "s"(p.N), "s"(p.K), "s"(p.kslice),
               "s"(p.tiles_m), "s"(p.tiles_n), "s"(p.tile_linear), "s"((int)gridDim.x), "s"((int)gridDim.y));
  // W8A8: the int8 weight rows are plain [N][K].  With the MFMA operand's own lane map (row = lane & 15, 16-B piece =
  // lane >> 4) a wave instruction touches 16 rows with every lane quad spanning four of them, and the L1 pulls 15 B/clk per
  // CU out of L2 (tools/l1_pattern_probe.hip) where this tile needs 32.  Row-coalesced (row = lane >> 2, piece = lane & 3:
  // a quad reads 64 consecutive bytes) the same loads run at 38 B/clk; the fragment is then turned into the operand map by
  // one ds_write_b128 + ds_read_b128 through a wave-private KiB, slot = piece * 16 + (row ^ 5 piece): the map among the
  // candidates of tools/lds_transpose_probe.hip whose write costs what a lane-linear one does (row ^ 4 piece: +23 %).
  constexpr bool W8C = MODE == MODE_W8 && OMNI_GEMM_W8_COALESCED;
  __shared__ __attribute__((aligned(16))) uint8_t wtr[W8C ? WAVES * 1024 : 16];
  __shared__ __attribute__((aligned(16))) uint32_t epi_w[TO_SLAB ? 4 : 64 * WAVES];   // {wscale, w_sz} per channel of the tile
  __shared__ uint32_t epi_a[TO_SLAB ? 4 : MT];                                          // {ascale, asum} per row of the tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  int tile_n = blockIdx.x, tile_m = blockIdx.z;
  if (p.tiles_n > 0) {
    // XCD-aware order (1-D grid).  Workgroups are dealt round-robin to the 8 XCDs, each with a private
    // 4 MiB L2: XCD x takes the x-th contiguous eighth of the tiles, enumerated as 8 x 8 super-blocks, so
    // the ~64 workgroups an XCD runs at a time share 8 activation row-tiles and 8 weight column-tiles
    // (every L2 line is fetched once from the fabric and reused 8x) instead of striding over all of M.
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
  const int ng = tile_n * WAVES + wave;  // 64-channel group
  const bool wave_active = (ng * 64) < p.N;
  const int ngc = wave_active ? ng : (p.N / 64 - 1);   // inactive waves stream the last valid group again and drop it
  const int m0 = tile_m * MT;
  const int k_begin = blockIdx.y * p.kslice;
  const int k_end = min(p.K, k_begin + p.kslice);
  const int nsteps = (k_end - k_begin) / KSTEP;
  const int nchunks = (nsteps + STEPS - 1) / STEPS;

  // ---- weight addressing -------------------------------------------------------------
  // W4: lane -> (x = tile row, c = n3, e = k6); W8: lane -> (i = row in 16-block, g = k piece)
  const int lx = (lane >> 3) & 1, lc = lane & 7, le = lane >> 4;
  const uint8_t* wbase;
  if constexpr (W8C) {
    wbase = p.W + (size_t)(ngc * 64 + (lane >> 2)) * p.K + (lane & 3) * 16;
  } else if constexpr (MODE == MODE_W8) {
    wbase = p.W + (size_t)(ngc * 64 + (lane & 15)) * p.K + (lane >> 4) * 16;
  } else {
    wbase = p.W + ((size_t)(2 * ngc + lx) * (p.K / 32)) * 512 + (lc * 4 + le) * 16;
  }
  uint8_t* const wtr_w = wtr + (W8C ? wave * 1024 + ((lane & 3) * 16 + ((lane >> 2) ^ (5 * (lane & 3)))) * 16 : 0);
  const uint8_t* const wtr_r = wtr + (W8C ? wave * 1024 + ((lane & 48) + ((lane & 15) ^ (5 * (lane >> 4)))) * 16 : 0);
  auto load_w = [&](int k, int j) -> uint4 {
    // W4: j = tile parity inside the 64-k step.  W8: j = 16-row block (0..3).
    const uint8_t* ptr;
    if constexpr (MODE == MODE_W8) ptr = wbase + (size_t)j * 16 * p.K + k;
    else ptr = wbase + (size_t)(k / 32 + j) * 512;
    v4i v;
    if constexpr (NT) v = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(ptr));
    else v = *reinterpret_cast<const v4i*>(ptr);
    return make_uint4((uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]);
  };
  constexpr int WL = (MODE == MODE_W8) ? 4 : 2;  // weight loads per lane per k-step
  // weight prefetch ring: a whole chunk (4 steps) ahead for the int4 modes (32 VGPRs); W8A8 rows are twice the bytes
  // (64 VGPRs for a chunk pushed the kernel over 256 VGPRs: 100 B/lane of scratch), so it runs two steps ahead
  // (W8A8 on the 128-row tile: one step ahead -- with two the instantiation spilled 3 - 7 VGPRs to scratch; no product kernel may
  //  spill, tests/test_code_objects_cpu.py.  This is the ragged-shape fallback: the models' shapes run w4a8_gemm_exact_kernel)
  constexpr int WRING = MODE == MODE_CHN ? OMNI_GEMM_RING_CHN : ((MODE == MODE_W8 && MB == 8) ? OMNI_GEMM_RING_W8 : OMNI_GEMM_RING_OTHER);
  uint4 wq[WRING][WL];

  // ---- activation staging --------------------------------------------------------------
  // LDS image of a chunk: [k-step s][16-B slot q of the step's B operand][row m][16 B] ("plane" layout):
  // the 16 lanes of a ds_read_b128 lane group read 16 consecutive rows of one or two planes = 16 distinct
  // 16-B slots of the 256-B bank row (conflict-free; the row-major [s][m][64 B] image was 2-way).
  // Thread -> (row m, 16-B piece kk of the chunk row) is chosen so that the 32 lanes of a store group hit
  // 32 distinct banks: W4 (ds_write_b32 of dword e into plane e): 8 rows x 4 pieces of one k-step;
  // W8 (ds_write_b128 into plane kk&3): 8 rows per 8-lane group.
  static_assert(KCHUNK == 256 && (MT % 16) == 0, "piece mapping assumes 16 pieces per chunk row");
  uint4 areg[A_LOADS];
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
  auto load_a = [&](int chunk, bool plain) {
    const int kc = k_begin + chunk * KCHUNK;
#pragma unroll
    for (int j = 0; j < A_LOADS; ++j) {
      int m, kk;
      piece(j, m, kk);
      const int k = kc + kk * 16;
      if (plain) {   // whole tile in range (workgroup-uniform, hoisted out of the steady loop): nothing to predicate
        areg[j] = *reinterpret_cast<const uint4*>(p.A + (size_t)(m0 + m) * p.K + k);
        continue;
      }
      // branch-free (a branch around a load turns every later wait into vmcnt(0)): out-of-range pieces read a
      // valid address and are zeroed by a select
      const bool ok = tid + j * NTHREADS < MT * KCHUNK / 16 && (m0 + m) < p.M && k < k_end;
      const int mr = (m0 + m) < p.M ? (m0 + m) : (p.M - 1);
      const int kr = k < k_end ? k : k_begin;
      const uint4 v = *reinterpret_cast<const uint4*>(p.A + (size_t)mr * p.K + kr);
      areg[j] = ok ? v : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int j = 0; j < A_LOADS; ++j) {
      if (tid + j * NTHREADS >= MT * KCHUNK / 16) continue;
      int m, kk;
      piece(j, m, kk);
      const int ks = kk >> 2;
      if constexpr (MODE == MODE_W8) {
        // (component-wise copy: assigning the whole uint4 keeps `areg` in scratch memory)
        *reinterpret_cast<uint4*>(&lds[buf][((ks * 4 + (kk & 3)) * MT + m) * 16]) =
            make_uint4(areg[j].x, areg[j].y, areg[j].z, areg[j].w);
      } else {
        // kk = 16-byte piece of the row: tp = tile parity, d = k5; its dword e belongs to slot e of the
        // step's B operand, at byte tp*8 + d*4 of the slot
        const int tp = (kk >> 1) & 1, d = kk & 1;
        uint8_t* dst = &lds[buf][(ks * 4 * MT + m) * 16 + tp * 8 + d * 4];
        *reinterpret_cast<uint32_t*>(dst + 0 * MT * 16) = areg[j].x;
        *reinterpret_cast<uint32_t*>(dst + 1 * MT * 16) = areg[j].y;
        *reinterpret_cast<uint32_t*>(dst + 2 * MT * 16) = areg[j].z;
        *reinterpret_cast<uint32_t*>(dst + 3 * MT * 16) = areg[j].w;
      }
    }
  };

  v4i acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) acc[mb][ab] = (v4i){0, 0, 0, 0};

  // ---- prologue --------------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < WRING; ++s) {
    const int ks = s < nsteps ? s : 0;      // (short slices) re-read step 0, never consumed
#pragma unroll
    for (int j = 0; j < WL; ++j) wq[s][j] = load_w(k_begin + ks * KSTEP, j);
  }
  load_a(0, false);
  // Epilogue operands -> LDS (published by the first chunk's barrier): the write-back reads them with LDS latency.  Round 2
  // read them from global memory inside the write-back loop, per row block: 32 dependent memory round trips per wave = the
  // time of ~5 K chunks per tile (K = 4096 ran at 0.44 of the int8 peak where K = 14336 reached 0.55).
  if constexpr (!TO_SLAB) {
    for (int i = tid; i < 64 * WAVES; i += NTHREADS) {
      const int n = min(tile_n * 64 * WAVES + i, p.N - 1);
      const uint32_t sw = __builtin_bit_cast(uint16_t, p.wscales[n]);
      uint32_t sz = 0;
      if constexpr (MODE == MODE_CHN) sz = __builtin_bit_cast(uint16_t, p.wsz[n]);
      epi_w[i] = sw | (sz << 16);
    }
    for (int i = tid; i < MT; i += NTHREADS) {
      const int m = min(m0 + i, p.M - 1);
      const uint32_t sa = __builtin_bit_cast(uint16_t, p.ascales[m]);
      uint32_t as = 0;
      if constexpr (MODE == MODE_CHN) as = __builtin_bit_cast(uint16_t, p.asum[m]);
      epi_a[i] = sa | (as << 16);
    }
  }
  store_a(0);

  // per-group second-level params for the current chunk (2 groups of 128 per chunk)
  uint32_t gs[2] = {0, 0}, gz[2] = {0, 0};
  if constexpr (MODE == MODE_GRP) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kg = (k_begin + h * 128 < k_end) ? k_begin + h * 128 : k_begin;
      gs[h] = *reinterpret_cast<const uint32_t*>(p.s2s + (size_t)(kg / 128) * p.N + ((size_t)(2 * ngc + lx) * 32 + lc * 4));
      gz[h] = *reinterpret_cast<const uint32_t*>(p.s2z + (size_t)(kg / 128) * p.N + ((size_t)(2 * ngc + lx) * 32 + lc * 4));
    }
  }

#ifdef OMNI_DEBUG_CLOCKS
  unsigned long long dbg_barrier = 0, dbg_vmwait = 0;
  const unsigned long long dbg_start = wall_clock64();
#endif
  // One K chunk.  STEADY = this chunk and the next are whole: no control flow at all, so the waits the compiler
  // inserts are counted (vmcnt(N)) and the weight refills / next activation tile stay in flight across the MFMAs.
  // The generic form (last chunks, ragged K) keeps the conditions.
  const size_t gcol = (size_t)(2 * ngc + lx) * 32 + lc * 4;
  auto run_chunk = [&](int c, auto steady_tag) {
    constexpr bool STEADY = decltype(steady_tag)::value;
    const int kc = k_begin + c * KCHUNK;
    const int steps_here = STEADY ? STEPS : min(STEPS, nsteps - c * STEPS);
    const bool has_next = STEADY ? true : (c + 1) < nchunks;
    if (has_next) load_a(c + 1, STEADY);
    uint32_t gsn[2] = {0, 0}, gzn[2] = {0, 0};   // second-level params of the NEXT chunk (consumed a chunk later)
    if constexpr (MODE == MODE_GRP) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kn = kc + KCHUNK + h * 128;
        const int kg = (STEADY || kn < k_end) ? kn : k_begin;
        gsn[h] = *reinterpret_cast<const uint32_t*>(p.s2s + (size_t)(kg / 128) * p.N + gcol);
        gzn[h] = *reinterpret_cast<const uint32_t*>(p.s2z + (size_t)(kg / 128) * p.N + gcol);
      }
    }
#ifdef OMNI_DEBUG_CLOCKS
    const unsigned long long dbg_t0 = wall_clock64();
#endif
    __syncthreads();  // chunk c of A is visible in lds[c&1]
#ifdef OMNI_DEBUG_CLOCKS
    dbg_barrier += wall_clock64() - dbg_t0;
#endif
    const uint8_t* abuf = lds[c & 1];

#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (STEADY || s < steps_here) {
        v4i wa[4];
        if constexpr (W8C) {
          // (issuing the transpose of step s + 1 behind the MFMAs of step s instead was 5-9 % slower: 16 more live
          // registers spill, and the fragment then has to be back from L2 one step earlier)
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
          // dwords of a chunk: x=(k5=0,n2=0) y=(k5=0,n2=1) z=(k5=1,n2=0) w=(k5=1,n2=1)
          const uint4 t0 = wq[s % WRING][0], t1 = wq[s % WRING][1];
          const uint32_t d[2][4] = {{t0.x, t0.z, t1.x, t1.z}, {t0.y, t0.w, t1.y, t1.w}};
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              uint32_t u[4];
#if defined(OMNI_GEMM_ABLATE) && (OMNI_GEMM_ABLATE & 1)   // timing experiment: no unpack (wrong results)
#pragma unroll
              for (int q = 0; q < 4; ++q) u[q] = d[b][q];
#else
#pragma unroll
              for (int q = 0; q < 4; ++q) u[q] = (d[b][q] >> (4 * a)) & 0x0F0F0F0Fu;
#endif
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
        // refill this step's weight registers with the step WRING ahead (the next chunk's for the int4 modes)
        if (STEADY || (c * STEPS + s + WRING < nsteps)) {
#pragma unroll
          for (int j = 0; j < WL; ++j) wq[s % WRING][j] = load_w(kc + (s + WRING) * KSTEP, j);
          // keep the refill HERE (WRING steps of MFMAs ahead of its use): the scheduler otherwise sinks all
          // the steps' loads to the end of the chunk, right in front of the wait that needs them
          if constexpr (STEADY) __builtin_amdgcn_sched_barrier(0x78F);   // everything but VMEM may still move across
        }
        // B operands (activation rows) from LDS, pinned three row blocks ahead of the MFMAs that use them: left alone
        // the compiler issues two reads and waits for them right away (read, read, lgkmcnt, 8 MFMAs, ...), i.e. one
        // exposed LDS latency per 8 MFMAs -- the MFMA pipe sat at 40-50 % busy (PMC) on exactly that
        v4i bf[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#if defined(OMNI_GEMM_ABLATE) && (OMNI_GEMM_ABLATE & 2)   // timing experiment: one LDS read per step (wrong results)
          bf[mb] = *reinterpret_cast<const v4i*>(abuf + ((s * 4 + (lane >> 4)) * MT + (lane & 15)) * 16);
#else
          bf[mb] = *reinterpret_cast<const v4i*>(abuf + ((s * 4 + (lane >> 4)) * MT + mb * 16 + (lane & 15)) * 16);
#endif
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int ab = 0; ab < 4; ++ab)
            acc[mb][ab] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[ab], bf[mb], acc[mb][ab], 0, 0, 0);
        // publish the next activation tile from INSIDE the chunk (its global loads were issued at the chunk's start,
        // OMNI_GEMM_STORE_A_STEP steps of MFMAs ago): at the chunk's end the ds_write latency, the barrier and the first
        // ds_read latency of the next chunk queued up back to back with no MFMA to cover them (PMC: 29 % of the wave
        // cycles parked on counters / barriers)
        if constexpr (STEADY) {
          if (s == OMNI_GEMM_STORE_A_STEP) store_a((c + 1) & 1);
        }
      }
    }
    if constexpr (STEADY && OMNI_GEMM_PIPE_B && MODE == MODE_CHN) {   // one read pipeline over the whole chunk (it also spans the step seams).  Per-channel mode only: per-group has no registers left
    // for it and W8A8 measured 3 % slower with it.  Gain: +2 % at 4096^3, +4 % on the K = 14336 shape, -1 % on gate_up
      constexpr int PRE = MB < 3 ? MB : 3;
      __builtin_amdgcn_sched_group_barrier(0x100, PRE, 0);
#pragma unroll
      for (int i = 0; i < STEPS * MB - PRE; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * PRE, 0);
    }
#ifdef OMNI_DEBUG_CLOCKS
    const unsigned long long dbg_t1 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (debug build) isolate the wait for the staged loads
    dbg_vmwait += wall_clock64() - dbg_t1;
#endif
    if (has_next && !(STEADY && OMNI_GEMM_STORE_A_STEP < STEPS)) store_a((c + 1) & 1);
    if constexpr (MODE == MODE_GRP) {
#pragma unroll
      for (int h = 0; h < 2; ++h) { gs[h] = gsn[h]; gz[h] = gzn[h]; }
    }
  };
  const int nfull = nsteps / STEPS;     // whole chunks
  int c = 0;
  // (per-group mode keeps the generic loop: its dequant temporaries + the pinned refills exceed 256 VGPRs and spill)
  if constexpr (MODE != MODE_GRP || OMNI_GEMM_GRP_STEADY) {
    if (m0 + MT <= p.M && MT * KCHUNK / 16 == A_LOADS * NTHREADS)   // all activation rows of the tile exist
      for (; c + 1 < nfull; ++c) run_chunk(c, BoolTag<true>{});
  }
  for (; c < nchunks; ++c) run_chunk(c, BoolTag<false>{});

#ifdef OMNI_DEBUG_CLOCKS
  if (blockIdx.x == 0 && (tid & 63) == 0) {   // per wave of workgroup 0: total / barrier / vmcnt wait (10 ns ticks)
    omni_dbg_clk[wave * 4 + 0] = wall_clock64() - dbg_start;
    omni_dbg_clk[wave * 4 + 1] = dbg_barrier;
    omni_dbg_clk[wave * 4 + 2] = dbg_vmwait;
  }
#endif
  if (!wave_active) return;

  // ---- write back ------------------------------------------------------------------------
  // D layout (16x16): col = lane&15 -> m, row = (lane>>4)*4 + r -> channel slot i.
  // W4: i = x*8 + c, channel = ng*64 + x*32 + ab*8 + c  (4 consecutive channels per lane)
  // W8: channel = ng*64 + rb*16 + i
  const int mcol = lane & 15;
  const int i0 = (lane >> 4) * 4;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = m0 + mb * 16 + mcol;
    if (m >= p.M) continue;
    float sa = 0.f, as = 0.f;
    if constexpr (!TO_SLAB) {
      const uint32_t av = epi_a[mb * 16 + mcol];
      sa = (float)__builtin_bit_cast(half_t, (uint16_t)(av & 0xFFFFu));
      as = (float)__builtin_bit_cast(half_t, (uint16_t)(av >> 16));
    }
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
      int n, nl;     // channel, and its index inside the workgroup's tile
      if constexpr (MODE == MODE_W8) nl = wave * 64 + ab * 16 + i0;
      else nl = wave * 64 + (i0 >> 3) * 32 + ab * 8 + (i0 & 7);
      n = tile_n * 64 * WAVES + nl;
      const v4i a4 = acc[mb][ab];
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
        *reinterpret_cast<uint2*>(p.out + (size_t)m * p.out_stride + n) =
            *reinterpret_cast<const uint2*>(o);
      }
    }
  }
}

}  // namespace omni
#include "qgemm_exact.h"
#include "qgemm_midm.h"
namespace omni {

// Rider workgroup of the fp16-input GEMV (A16): activation row blockIdx.x.  Replays invoke_quant_fuse_sum's row sum --
// thread t of min(K, 1024) virtual threads adds x[t], x[t + nv], ... in f32, then the 32-lane / 32-warp butterflies
// (fused_kernels.cu:108-127, reduction_utils.cuh:47-85) -- with the machinery the row kernels use (row_reduce.h), so the
// fp16 sum equals omni_quant_fuse_sum's bit for bit; scale = h(amax / 127) from the producer's row maximum.
template <int NTHR>
__device__ __forceinline__ void a16_rider(const GemmArgs& p, float* xs, float* red) {
  const int row = blockIdx.x;
  if (row >= p.M) return;
  const int hidden = p.K;
  const int nv = hidden < 1024 ? hidden : 1024;
  const int t = threadIdx.x;
  const half_t* x = p.A16 + (size_t)row * p.K;
  const float rowmax = amax_rows_wave(p.amax);      // (every wave reads all 16 rows; lane `row` of each holds this row's)
  if (!p.sum_out) {       // nobody reads a row sum downstream (per-group, W8A8: no zero-point term): the scale alone
    if (t == row) p.scale_out[row] = (half_t)(rowmax / 127.0f);
    return;
  }
  for (int i = t * VT; i < hidden; i += NTHR * VT) {
    const v8h v = *reinterpret_cast<const v8h*>(x + i);
    *reinterpret_cast<v4f*>(xs + i) = (v4f){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    *reinterpret_cast<v4f*>(xs + i + 4) = (v4f){(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
  }
  __syncthreads();
  float s[1][VT], tot[1];
  ordered_partials<1>(xs, t, nv, hidden, s, [](float (&v)[1][VT], int e, float val) { v[0][e] = v[0][e] + val; });
  tree_sum8<1>(s, red, t, nv >> 5, tot);
  if (t == row) {                                   // (row < 16 <= 64: a lane of wave 0)
    if (p.sum_out) p.sum_out[row] = (half_t)tot[0];
    p.scale_out[row] = (half_t)(rowmax / 127.0f);
  }
}

// ------------------------------------------------------------------------------------------
// Decode-shape kernel (M <= 128): pure weight streaming.
//   * each wave keeps a ring of RING k-steps (RING x 2 KiB) of packed weights in flight: the
//     registers of step s are refilled for step s+RING right after they are unpacked, so the HBM
//     latency is covered by RING-1 steps of MFMA work;
//   * the int8 activations are staged one ROUND (RING steps) ahead: global/L2 -> VGPR at the top
//     of round r, -> LDS (in the weights' k order, double buffered) at its end.  On-GPU ablation
//     (tools/gemv_probe.hip, profiles/) showed that staging the whole K-slice up front costs up to
//     2x on this kernel, while the stream + unpack + LDS reads + MFMA alone run at the plain-copy rate;
//   * single-wave tiles (finest scheduling granule, no barrier inside the K loop) of 16 / 32 / 64 rows; M = 65..128 is
//     two 64-row tiles per channel group (grid.z).
//   * no control flow inside a round, so every wait is a counted s_waitcnt vmcnt(N).
// ------------------------------------------------------------------------------------------
#ifndef OMNI_GEMV_RING_MB4
#define OMNI_GEMV_RING_MB4 4
#endif
#ifndef OMNI_GEMV_AR_MB4
#define OMNI_GEMV_AR_MB4 2      // k-steps per activation round of the 64-row tile: 2 = 8-KiB rounds, four K parts per
#endif                          // workgroup, two waves per SIMD (measured 10-15 % faster at M = 64 than 4 = 16-KiB rounds, two parts)
#ifndef OMNI_GEMV_FENCE_REFILL
#define OMNI_GEMV_FENCE_REFILL 1
#endif
#ifndef OMNI_GEMV_ABLATE
#define OMNI_GEMV_ABLATE 0      // timing experiments (wrong results): 1 no activation reloads, 2 a quarter of the MFMAs,
#endif                          // 4 no weight reloads, 8 no LDS publication of the next activation round
// VAR = 1 (MB = 2, int4 modes): the 32-row tile in the geometry used for M = 33..128 where it beats the 64-row tile --
// 16-KiB activation rounds of 4 k-steps, a 4-step weight ring, four K parts per workgroup (profiles/r03_c_*).
template <int MB, int MODE, int VAR = 0>
struct GemvCfg {
  static constexpr bool NARROW = VAR == 1 && MB == 2 && MODE != MODE_W8;
  static constexpr int WL = (MODE == MODE_W8) ? 4 : 2;
  // AR = k-steps per activation round (one LDS buffer), RING = k-steps of weights in flight per wave (a multiple of AR:
  // the 64-row tile keeps 16-KiB activation rounds but a deeper weight ring -- bytes in flight per wave are what bounds it)
  static constexpr int AR = (MODE == MODE_W8) ? 4 : (NARROW ? 4 : (MB <= 2 ? 8 : OMNI_GEMV_AR_MB4));
  static constexpr int RING = (MODE == MODE_W8) ? 4 : (NARROW ? 4 : (MB <= 2 ? 8 : OMNI_GEMV_RING_MB4));
  // single-wave tiles of up to 64 rows (no barrier; K is split over the workgroup's KW waves instead).  M = 65..128
  // runs as two 64-row tiles per channel group (grid.z = 2): the second read of the packed weights is served by
  // L2 / MALL.  (Round 1 had a 128-row tile here -- four channel groups per workgroup sharing a 32-KiB activation
  // round, 128 accumulator registers per wave spilling into AGPRs, an 8-KiB weight ring: 0.6 TB/s; removed.)
  static constexpr int WAVES = 1;
  static constexpr int MAX_KW = (MB == 1 || NARROW || (MB == 4 && MODE != MODE_W8 && OMNI_GEMV_AR_MB4 == 2)) ? 4 : 2;   // LDS: KW x 2 buffers x MT x RK <= 64 KiB
};

//   * KW > 1 (single 64-channel group per workgroup only): KW waves split the workgroup's K-slice,
//     each streaming its own part with its own LDS buffers and no barrier; the int32 partials meet in
//     LDS at the end and wave w finishes the 4/KW row blocks it owns.  More waves in flight per CU
//     without the slab round trip of a grid-level split.
// NT: weight loads carry the non-temporal hint (weights streamed once from HBM, the default); false = plain loads,
// for weights a preceding row kernel has prefetched into the L2s (omni_prefetch_arm_gemm; take_prefetched_weight below).
// MZ = 2 (M = 65..128): the workgroup carries BOTH 64-row tiles of its channel group -- waves (kw, half) -- so the two
// reads of a weight byte are issued side by side by waves of one CU (the second is served by L1 / the in-flight line in
// L2) instead of by two workgroups somewhere on the chip at different times.
// EPI = 1 (gate_up projection, fused extension): the wave's two tile rows are tile row g of the GATE half and tile row g
// of the UP half (instead of two consecutive ones), so lane l < 32 finishes gate channel c of row m and lane l + 32 the
// matching up channel: the epilogue rounds both to fp16 exactly as the plain kernel stores them, exchanges them across
// the wave halves, applies silu_and_mul's arithmetic and stores act[M, N/2] -- the fp16 tensor silu_and_mul would have
// written -- plus the row maxima of |act| (integer atomicMax on the f32 bits: exact and order independent).
// A16 = true (o / down projection, fused extension): the int8 codes are produced on the fly from fp16 activations and
// the row maxima the producer left in `amax` (code = rni_sat(x * (127 / amax)), invoke_quant's arithmetic,
// fused_kernels.cu:126-131); the first grid row (blockIdx.y == 0) are RIDER workgroups, one per activation
// row, which replay the reference's ordered row sum (fused_kernels.cu:108-127) and write sum / scale for the consumer
// of the slabs.  Together they remove the quant row kernel between two GEMVs, bit for bit.
template <int MB, int MODE, bool TO_SLAB, int KW = 1, bool NT = true, int MZ = 1, int EPI = 0, bool A16 = false, int VAR = 0>
__global__ __launch_bounds__((64 * GemvCfg<MB, MODE, VAR>::WAVES * KW * MZ), (((MB == 4 && KW == 4 && MZ == 1) || A16 || VAR == 1) ? 2 : 1)) void w4a8_gemv_kernel(GemmArgs p) {
  constexpr int MT = MB * 16;
  static_assert(EPI == 0 || (!TO_SLAB && MZ == 1), "SiLU epilogue: in-kernel epilogue, one row tile");
  static_assert(!A16 || (TO_SLAB && MZ == 1 && MB == 1 && GemvCfg<MB, MODE, VAR>::WAVES == 1), "fp16-input form: slab output, one 16-row single-wave tile");
  static_assert(EPI == 0 || MB == 1, "row-maximum hand-off covers 16 rows");
  constexpr int WAVES = GemvCfg<MB, MODE, VAR>::WAVES;
  static_assert(KW == 1 || WAVES == 1, "in-workgroup K split is for single-wave tiles");
  static_assert(KW <= GemvCfg<MB, MODE, VAR>::MAX_KW, "LDS budget");
  static_assert(KW == 1 || KW == 2 || KW == 4, "KW");
  static_assert(MZ == 1 || (MZ == 2 && WAVES == 1), "row-tile pairs are for single-wave tiles");
  constexpr int NTHREADS = 64 * WAVES;                   // threads sharing one staged activation tile
  constexpr int WL = GemvCfg<MB, MODE, VAR>::WL;
  constexpr int RING = GemvCfg<MB, MODE, VAR>::RING;
  constexpr int AR = GemvCfg<MB, MODE, VAR>::AR;
  constexpr int SUB = RING / AR;                         // activation rounds per ring round
  static_assert(RING % AR == 0 && (SUB == 1 || SUB % 2 == 0), "ring = whole activation rounds, buffer parity static");
  constexpr int RK = AR * KSTEP;                         // k per activation round
  constexpr int APT = (MT * RK / 16) / NTHREADS;          // 16-B activation pieces per thread per round
  static_assert((MT * RK / 16) % NTHREADS == 0, "activation round must tile the workgroup");
  __shared__ __attribute__((aligned(16))) uint8_t lds_all[KW * MZ][2][MT * RK];
  __shared__ float epi_red[A16 ? 96 : 1];   // A16 rider: scratch of the reduction tree

  // One batch of scalar loads for the kernel arguments in front of the weight stream: hipcc otherwise requests the pointers
  // behind its first branch on N (the SiLU form: a second, dependent kernarg round trip in front of the first weight load).
  if constexpr (A16) asm volatile("" ::"s"(p.A16), "s"(p.W), "s"(p.amax), "s"(p.N), "s"(p.K), "s"(p.kslice), "s"(p.M));
  else asm volatile("" ::"s"(p.A), "s"(p.W), "s"(p.wscales), "s"(p.ascales), "s"(p.wsz), "s"(p.asum), "s"(p.N), "s"(p.K), "s"(p.kslice), "s"(p.M));
  if constexpr (A16) {
    if (blockIdx.y == 0) {      // rider workgroups (dispatched first): ordered row sum + scale of activation row blockIdx.x
      if (!OMNI_GEMV_DBG_BIT(p, 1)) a16_rider<64 * KW>(p, reinterpret_cast<float*>(&lds_all[0][0][0]), epi_red);
      return;
    }
  }
  const int by = A16 ? (int)blockIdx.y - 1 : (int)blockIdx.y;     // K slice of this workgroup
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int half = MZ > 1 ? wave / KW : 0;                // row tile of this wave inside the workgroup
  const int kw = (KW > 1 || MZ > 1) ? wave % KW : 0;      // K part of this wave
  const int tid = (KW > 1 || MZ > 1) ? lane : threadIdx.x;   // index inside the staging group
  uint8_t (*lds)[MT * RK] = lds_all[kw + KW * half];
  const int ng = (KW > 1 || MZ > 1) ? blockIdx.x : blockIdx.x * WAVES + wave;
  const int m0 = (blockIdx.z * MZ + half) * MT;          // row tile (M > 64: grid.z = 2, or both tiles in the workgroup)
  const bool wave_active = (ng * 64) < p.N;
  const int kpart = p.kslice / KW;                        // host: kslice % (64 * KW) == 0 when KW > 1
  const int k_begin = by * p.kslice + kw * kpart;
  const int k_end = min(p.K, k_begin + kpart);
  const int nsteps = (k_end - k_begin) / KSTEP;
  const int rounds = nsteps / RING;

  const int lx = (lane >> 3) & 1, lc = lane & 7, le = lane >> 4;
  // inactive waves (N/64 not a multiple of WAVES) stream the last valid group again and drop it
  const int ngc = wave_active ? ng : (p.N / 64 - 1);
  const uint8_t* wbase;
  // (W8A8: the operand lane map is kept here -- the row-coalesced fetch + LDS transpose of w4a8_gemm_kernel measured no
  //  gain at M = 1: this kernel waits on HBM latency, not on the L1's 15 B/clk)
  // tile row (32 channels) this lane streams: two consecutive rows of the 64-channel group, or (EPI = 1) row ngc of the
  // gate half and row ngc of the up half
  const int trow = EPI == 1 ? (lx ? p.N / 64 + ngc : ngc) : 2 * ngc + lx;
  // (W8A8, EPI = 1: 16-row block j of the wave = gate channels ngc * 32 + 8 j + (0..7) in operand rows 0..7 and the
  //  matching up channels in rows 8..15, so that -- as in the int4 layout -- lanes < 32 finish gate outputs and lanes
  //  >= 32 the up outputs of the same (row, 4 channels))
  if constexpr (MODE == MODE_W8 && EPI == 1)
    wbase = p.W + (size_t)(((lane & 8) ? p.N / 2 : 0) + ngc * 32 + (lane & 7)) * p.K + (lane >> 4) * 16;
  else if constexpr (MODE == MODE_W8) wbase = p.W + (size_t)(ngc * 64 + (lane & 15)) * p.K + (lane >> 4) * 16;
  else wbase = p.W + ((size_t)trow * (p.K / 32)) * 512 + (lc * 4 + le) * 16;
  auto load_w = [&](int k, int j) -> uint4 {
    const uint8_t* ptr;
    if constexpr (MODE == MODE_W8) ptr = wbase + (size_t)j * (EPI == 1 ? 8 : 16) * p.K + k;
    else ptr = wbase + (size_t)(k / 32 + j) * 512;
    v4i v;
    if constexpr (NT) v = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(ptr));
    else v = *reinterpret_cast<const v4i*>(ptr);
    return make_uint4((uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]);
  };
  const size_t gcol = (size_t)trow * 32 + lc * 4;  // per-group param column of this lane
  auto load_gp = [&](const uint8_t* base, int k) -> uint32_t {
    return *reinterpret_cast<const uint32_t*>(base + (size_t)(k / 128) * p.N + gcol);
  };

  // activation staging: piece id -> (row m, 16-byte piece kk of the round).  A single-wave tile (MB = 1,
  // RING = 8) lets every ds_write cover 4 rows x 16 pieces (4 k-steps): with the rotated piece position
  // below that is one lane per LDS bank.
  constexpr bool QUAD_ROWS = (NTHREADS == 64 && (RK / 16) % 16 == 0 && MODE != MODE_W8);
  constexpr int CPR = RK / 256;                          // 16-piece chunks per activation row of a round
  auto piece = [&](int j, int& m, int& kk) {
    if constexpr (QUAD_ROWS) {
      m = 4 * (j / CPR) + (tid >> 4);
      kk = (tid & 15) + 16 * (j % CPR);
    } else {
      const int id = tid + j * NTHREADS;
      m = id / (RK / 16);
      kk = id % (RK / 16);
    }
  };
  uint4 areg[A16 ? 1 : APT];
  uint4 araw[A16 ? APT : 1][2];      // A16: the 16 fp16 values behind each 16-code piece
  float qrow[A16 ? APT : 1];         // A16: 127 / amax of the row of piece j (rows do not change from round to round)
  float qlane = 0.0f;                // A16: 127 / amax of row (lane & 15)
  AmaxRaw amax_raw;                  // A16: requested at kernel start, consumed after the weight ring is in flight
  if constexpr (A16) amax_raw = amax_rows_issue(p.amax);
  auto compute_q = [&]() {
    // (pins the reduction of the candidates HERE: called from two places, the optimiser otherwise hoists the common
    //  code -- and the wait for these loads -- to the top of the kernel, in front of every other request)
    asm volatile("" : "+v"(amax_raw.a0.x), "+v"(amax_raw.a0.y), "+v"(amax_raw.a0.z), "+v"(amax_raw.a0.w),
                      "+v"(amax_raw.a1.x), "+v"(amax_raw.a1.y), "+v"(amax_raw.a1.z), "+v"(amax_raw.a1.w),
                      "+v"(amax_raw.b0.x), "+v"(amax_raw.b0.y), "+v"(amax_raw.b0.z), "+v"(amax_raw.b0.w),
                      "+v"(amax_raw.b1.x), "+v"(amax_raw.b1.y), "+v"(amax_raw.b1.z), "+v"(amax_raw.b1.w));
    qlane = quant_multiplier(amax_rows_finish(amax_raw));
#pragma unroll
    for (int j = 0; j < APT; ++j) {
      if (QUAD_ROWS && (j % CPR) != 0) { qrow[j] = qrow[j > 0 ? j - 1 : 0]; continue; }   // same row as the previous piece
      int m, kk;
      piece(j, m, kk);
      const int mc = m0 + m < p.M ? m0 + m : p.M - 1;
      qrow[j] = __shfl(qlane, mc & 15, 64);
    }
  };
  auto load_a = [&](int kr) {  // kr = first k of the round
#pragma unroll
    for (int j = 0; j < APT; ++j) {
      int m, kk;
      piece(j, m, kk);
      const int mc = m0 + m < p.M ? m0 + m : p.M - 1;  // rows >= M re-read the last row (results never stored)
      if constexpr (A16) {
        const half_t* src = p.A16 + (size_t)mc * p.K + kr + kk * 16;
        araw[j][0] = *reinterpret_cast<const uint4*>(src);
        araw[j][1] = *reinterpret_cast<const uint4*>(src + 8);
      } else {
        areg[j] = *reinterpret_cast<const uint4*>(p.A + (size_t)mc * p.K + kr + kk * 16);
      }
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int j = 0; j < APT; ++j) {
      int m, kk;
      piece(j, m, kk);
      uint4 av;
      if constexpr (A16) {
        if (OMNI_GEMV_DBG_BIT(p, 2)) av = make_uint4(araw[j][0].x ^ araw[j][1].x, araw[j][0].y ^ araw[j][1].y, araw[j][0].z ^ araw[j][1].z, araw[j][0].w ^ araw[j][1].w);
        else
        av = make_uint4(quant4_f16(araw[j][0].x, araw[j][0].y, qrow[j]), quant4_f16(araw[j][0].z, araw[j][0].w, qrow[j]),
                        quant4_f16(araw[j][1].x, araw[j][1].y, qrow[j]), quant4_f16(araw[j][1].z, araw[j][1].w, qrow[j]));
      } else {
        av = make_uint4(areg[j].x, areg[j].y, areg[j].z, areg[j].w);
      }
      if constexpr (MODE == MODE_W8) {
        *reinterpret_cast<uint4*>(&lds[buf][((kk >> 2) * MT + m) * 64 + (kk & 3) * 16]) =
            make_uint4(av.x, av.y, av.z, av.w);   // component-wise: keeps the staging registers in VGPRs
      } else {
        // dword e of the piece goes to 16-B slot (e + kp) & 3 of row (kp, m): see w4a8_gemm_kernel
        const int kp = kk >> 2, tp = (kk >> 1) & 1, d = kk & 1;
        uint8_t* dst = &lds[buf][(kp * MT + m) * 64 + tp * 8 + d * 4];
        *reinterpret_cast<uint32_t*>(dst + ((0 + kp) & 3) * 16) = av.x;
        *reinterpret_cast<uint32_t*>(dst + ((1 + kp) & 3) * 16) = av.y;
        *reinterpret_cast<uint32_t*>(dst + ((2 + kp) & 3) * 16) = av.z;
        *reinterpret_cast<uint32_t*>(dst + ((3 + kp) & 3) * 16) = av.w;
      }
    }
  };

  v4i acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) acc[mb][ab] = (v4i){0, 0, 0, 0};

  // epilogue operands of the row blocks this wave finishes: requested now, so the kernel does not end
  // on a dependent HBM round trip
  constexpr int ABW = 4 / KW;
  const int ab0 = kw * ABW;
  const int mcol = lane & 15;
  const int i0 = (lane >> 4) * 4;
  auto chan = [&](int ab) -> int {
    if constexpr (EPI == 1) return (i0 >> 3) * (p.N / 2) + ng * 32 + ab * 8 + (i0 & 7);   // gate | up channel
    else if constexpr (MODE == MODE_W8) return ng * 64 + ab * 16 + i0;
    else return ng * 64 + (i0 >> 3) * 32 + ab * 8 + (i0 & 7);
  };
  uint2 swv[ABW], szv[ABW];
  half_t sav[MB], asv[MB];
  // 64-row tiles with the in-kernel epilogue: the operands (2 ABW + MB dwords per lane) are parked in LDS across the K loop --
  // held in registers they pushed these instantiations to 4 - 10 spilled VGPRs (scratch traffic inside the loop); the park
  // happens where the prologue waits for its first activation round anyway (loads return in order)
  constexpr bool EPI_PARK = OMNI_GEMV_EPI_PARK && MB == 4 && !TO_SLAB && KW == 4;
  constexpr int EPI_WORDS = 8;
  static_assert(!EPI_PARK || (4 * ABW + MB <= EPI_WORDS), "parked epilogue operands");
  __shared__ __attribute__((aligned(16))) uint32_t epi_park[EPI_PARK ? KW * MZ * 64 * EPI_WORDS : 4];
  uint32_t* const my_park = epi_park + (EPI_PARK ? (threadIdx.x * EPI_WORDS) : 0);
  auto park_epi = [&]() {
    if constexpr (EPI_PARK) {
      uint32_t w[EPI_WORDS] = {swv[0].x, swv[0].y, 0u, 0u, 0u, 0u, 0u, 0u};
      if constexpr (MODE == MODE_CHN) { w[2] = szv[0].x; w[3] = szv[0].y; }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        uint32_t v = __builtin_bit_cast(uint16_t, sav[mb]);
        if constexpr (MODE == MODE_CHN) v |= (uint32_t)__builtin_bit_cast(uint16_t, asv[mb]) << 16;
        w[4 + mb] = v;
      }
      *reinterpret_cast<uint4*>(my_park) = make_uint4(w[0], w[1], w[2], w[3]);
      *reinterpret_cast<uint4*>(my_park + 4) = make_uint4(w[4], w[5], w[6], w[7]);
    }
  };
  auto unpark_epi = [&]() {
    if constexpr (EPI_PARK) {
      const uint4 lo = *reinterpret_cast<const uint4*>(my_park), hi = *reinterpret_cast<const uint4*>(my_park + 4);
      swv[0] = make_uint2(lo.x, lo.y);
      if constexpr (MODE == MODE_CHN) szv[0] = make_uint2(lo.z, lo.w);
      const uint32_t h[4] = {hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        sav[mb] = __builtin_bit_cast(half_t, (uint16_t)(h[mb] & 0xFFFFu));
        if constexpr (MODE == MODE_CHN) asv[mb] = __builtin_bit_cast(half_t, (uint16_t)(h[mb] >> 16));
      }
    }
  };
  if constexpr (!TO_SLAB) {
#pragma unroll
    for (int j = 0; j < ABW; ++j) {
      const int n = wave_active ? chan(ab0 + j) : 0;
      swv[j] = *reinterpret_cast<const uint2*>(p.wscales + n);
      if constexpr (MODE == MODE_CHN) szv[j] = *reinterpret_cast<const uint2*>(p.wsz + n);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int m = m0 + mb * 16 + mcol;
      const int mc = m < p.M ? m : p.M - 1;
      sav[mb] = p.ascales[mc];
      if constexpr (MODE == MODE_CHN) asv[mb] = p.asum[mc];
    }
  }

  auto unpack = [&](const uint4 (&w)[WL], uint32_t sc4, uint32_t zr4, v4i (&wa)[4]) {
    if constexpr (MODE == MODE_W8) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) wa[rb] = (v4i){(int)w[rb].x, (int)w[rb].y, (int)w[rb].z, (int)w[rb].w};
    } else {
      const uint4 t0 = w[0], t1 = w[1];
      const uint32_t d[2][4] = {{t0.x, t0.z, t1.x, t1.z}, {t0.y, t0.w, t1.y, t1.w}};
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
  auto mma_step = [&](const v4i (&wa)[4], const uint8_t* abuf, int s) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int pos = (MODE == MODE_W8) ? (lane >> 4) : (((lane >> 4) + s) & 3);
      const v4i bf = *reinterpret_cast<const v4i*>(abuf + ((s * MT + mb * 16 + (lane & 15)) * 4 + pos) * 16);
#if (OMNI_GEMV_ABLATE & 2)      // timing experiment: one MFMA per row block instead of four (wrong results)
      acc[mb][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[0] ^ wa[1] ^ wa[2] ^ wa[3], bf, acc[mb][0], 0, 0, 0);
#else
#pragma unroll
      for (int ab = 0; ab < 4; ++ab)
        acc[mb][ab] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[ab], bf, acc[mb][ab], 0, 0, 0);
#endif
    }
  };

  uint4 wq[RING][WL];
  uint32_t gs[RING], gz[RING];
  if (rounds > 0) {
    // ---- prologue: ring for round 0 in flight, activations of round 0 into LDS ----------------
    // (A16: the fp16 activations are requested BEFORE the weights -- loads return in order, so they are back from L2
    //  while the ring is still in flight from HBM and the int8 conversion (~800 VALU per wave) runs under the weights'
    //  flight time; requested after the ring, it sat exposed behind the last weight byte: +1.5 us per launch)
    if constexpr (A16) load_a(k_begin);
#pragma unroll
    for (int s = 0; s < RING; ++s) {
#pragma unroll
      for (int j = 0; j < WL; ++j) wq[s][j] = load_w(k_begin + s * KSTEP, j);
      if constexpr (MODE == MODE_GRP) {
        gs[s] = load_gp(p.s2s, k_begin + s * KSTEP);
        gz[s] = load_gp(p.s2z, k_begin + s * KSTEP);
      }
    }
    if constexpr (!A16) load_a(k_begin);
    if constexpr (A16) {
      // (first wait of the kernel: the row maxima, requested before everything else.  The fence keeps the scheduler from
      //  hoisting their reduction -- and with it a vmcnt(0) -- in front of the activation and weight requests)
      __builtin_amdgcn_sched_barrier(0);
      compute_q();
    }
    store_a(0);
    park_epi();
    if constexpr (WAVES > 1) __syncthreads();
    // ---- steady state ---------------------------------------------------------------------------
    for (int r = 0; r + 1 < rounds; ++r) {
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) {
#if !(OMNI_GEMV_ABLATE & 1)
        load_a(k_begin + ((r * SUB + sub) + 1) * RK);
#endif
        const int par = SUB == 1 ? (r & 1) : (sub & 1);
        const uint8_t* abuf = lds[par];
#pragma unroll
        for (int s = 0; s < AR; ++s) {
          const int slot = sub * AR + s;
          v4i wa[4];
          unpack(wq[slot], gs[slot], gz[slot], wa);
          const int kn = k_begin + ((r + 1) * RING + slot) * KSTEP;
#if !(OMNI_GEMV_ABLATE & 4)
#pragma unroll
          for (int j = 0; j < WL; ++j) wq[slot][j] = load_w(kn, j);
#endif
          if constexpr (MODE == MODE_GRP) {
            gs[slot] = load_gp(p.s2s, kn);
            gz[slot] = load_gp(p.s2z, kn);
          }
          // 32-/64-row tiles: fence the refill in place.  Left alone hipcc sinks every refill of the round behind the
          // round's last MFMA -- the ISA then waits vmcnt(7..0) at the top of the next round for loads issued a few
          // cycles earlier: one exposed HBM round trip per round, covered only by the other wave of the SIMD (the M <= 16
          // kernel, eight steps per round, keeps its refills where they are written).
          if constexpr (OMNI_GEMV_FENCE_REFILL && MB > 1) __builtin_amdgcn_sched_barrier(0);
          mma_step(wa, abuf, s);
        }
#if !(OMNI_GEMV_ABLATE & 8)
        store_a(par ^ 1);
#endif
        if constexpr (WAVES > 1) __syncthreads();
      }
    }
    {  // last round: drain the ring
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) {
        if (sub + 1 < SUB) load_a(k_begin + (((rounds - 1) * SUB + sub) + 1) * RK);
        const int par = SUB == 1 ? ((rounds - 1) & 1) : (sub & 1);
        const uint8_t* abuf = lds[par];
#pragma unroll
        for (int s = 0; s < AR; ++s) {
          const int slot = sub * AR + s;
          v4i wa[4];
          unpack(wq[slot], gs[slot], gz[slot], wa);
          mma_step(wa, abuf, s);
        }
        if (sub + 1 < SUB) {
          store_a(par ^ 1);
          if constexpr (WAVES > 1) __syncthreads();
        }
      }
    }
  }
  // ---- leftovers (nsteps % RING): one step at a time, unpipelined (planner avoids this) ------------
  if constexpr (A16) {
    if (rounds == 0) compute_q();
  }
  if (rounds == 0) park_epi();
  for (int st = rounds * RING; st < nsteps; ++st) {
    const int kn = k_begin + st * KSTEP;
    if constexpr (WAVES > 1) __syncthreads();
    for (int id = tid; id < MT * 4; id += NTHREADS) {
      const int m = id >> 2, kk = id & 3;
      const int mc = m0 + m < p.M ? m0 + m : p.M - 1;
      uint4 a;
      if constexpr (A16) {
        const half_t* src = p.A16 + (size_t)mc * p.K + kn + kk * 16;
        const uint4 r0 = *reinterpret_cast<const uint4*>(src), r1 = *reinterpret_cast<const uint4*>(src + 8);
        const float q = __shfl(qlane, mc & 15, 64);
        a = make_uint4(quant4_f16(r0.x, r0.y, q), quant4_f16(r0.z, r0.w, q), quant4_f16(r1.x, r1.y, q), quant4_f16(r1.z, r1.w, q));
      } else {
        a = *reinterpret_cast<const uint4*>(p.A + (size_t)mc * p.K + kn + kk * 16);
      }
      if constexpr (MODE == MODE_W8) {
        *reinterpret_cast<uint4*>(&lds[0][m * 64 + kk * 16]) = a;
      } else {
        const int tp = (kk >> 1) & 1, d = kk & 1;
        uint8_t* dst = &lds[0][m * 64 + tp * 8 + d * 4];
        *reinterpret_cast<uint32_t*>(dst + 0) = a.x;
        *reinterpret_cast<uint32_t*>(dst + 16) = a.y;
        *reinterpret_cast<uint32_t*>(dst + 32) = a.z;
        *reinterpret_cast<uint32_t*>(dst + 48) = a.w;
      }
    }
    if constexpr (WAVES > 1) __syncthreads();
    uint4 w[WL];
#pragma unroll
    for (int j = 0; j < WL; ++j) w[j] = load_w(kn, j);
    uint32_t sc4 = 0, zr4 = 0;
    if constexpr (MODE == MODE_GRP) { sc4 = load_gp(p.s2s, kn); zr4 = load_gp(p.s2z, kn); }
    v4i wa[4];
    unpack(w, sc4, zr4, wa);
    mma_step(wa, lds[0], 0);
  }
  if (!wave_active) return;   // (never taken with KW > 1: one group per workgroup)

  // ---- combine the K parts (KW > 1) and write back (same mapping as w4a8_gemm_kernel) ----------------
  // (row blocks are indexed with compile-time constants only: a runtime index into acc[][] would push the
  //  accumulators to scratch memory)
  if constexpr (KW > 1) {
    // every wave used only its own buffers so far: park the partials there, then meet
    static_assert(MB * 4 * 64 * 16 <= 2 * MT * RK, "partials must fit the wave's staging buffers");
    v4i* mine = reinterpret_cast<v4i*>(&lds_all[kw + KW * half][0][0]);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) mine[(mb * 4 + ab) * 64 + lane] = acc[mb][ab];
    __syncthreads();
  }
  unpark_epi();
  float rowmax[MB];     // EPI = 1: max |act| of row (mb, lane & 15) over the channels this lane finished
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) rowmax[mb] = 0.0f;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = m0 + mb * 16 + mcol;
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
      if ((ab / ABW) != kw) continue;          // row block finished by another wave of the workgroup
      const int j = ab % ABW;                   // compile-time slot of the prefetched epilogue operands
      v4i a4 = acc[mb][ab];
      if constexpr (KW > 1) {
        a4 = (v4i){0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < KW; ++w)
          a4 += reinterpret_cast<const v4i*>(&lds_all[w + KW * half][0][0])[(mb * 4 + ab) * 64 + lane];
      }
      if (m >= p.M) continue;
      const int n = chan(ab);
      if constexpr (TO_SLAB) {
        int32_t* dst = p.slab + ((size_t)by * p.M + m) * p.N + n;
        *reinterpret_cast<v4i*>(dst) = a4;
      } else {
        typedef _Float16 v4h_t __attribute__((ext_vector_type(4)));
        const v4h_t sw4 = __builtin_bit_cast(v4h_t, swv[j]);
        v4h_t sz4 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
        if constexpr (MODE == MODE_CHN) sz4 = __builtin_bit_cast(v4h_t, szv[j]);
        const float sa = (float)sav[mb];
        float as = 0.f;
        if constexpr (MODE == MODE_CHN) as = (float)asv[mb];
        half_t o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = epilogue<MODE>(a4[r], (float)sw4[r], sa, (float)sz4[r], as);
        if constexpr (EPI == 1) {
          // lanes < 32 hold the fp16 gate outputs, lanes >= 32 the up outputs of the same (row, 4 channels)
          const uint2 mine = *reinterpret_cast<const uint2*>(o);
          const uint2 other = make_uint2((uint32_t)__shfl_xor((int)mine.x, 32, 64), (uint32_t)__shfl_xor((int)mine.y, 32, 64));
          const uint2 g2 = lane < 32 ? mine : other, u2 = lane < 32 ? other : mine;
          typedef _Float16 v4h_t2 __attribute__((ext_vector_type(4)));
          const v4h_t2 g4 = __builtin_bit_cast(v4h_t2, g2), u4 = __builtin_bit_cast(v4h_t2, u2);
          half_t act[4];
          float mx = 0.0f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            act[r] = silu_mul_h(g4[r], u4[r]);
            mx = __builtin_fmaxf(mx, __builtin_fabsf((float)act[r]));
          }
          rowmax[mb] = __builtin_fmaxf(rowmax[mb], mx);
          if (lane < 32)
            *reinterpret_cast<uint2*>(p.out + (size_t)m * p.out_stride + (ng * 32 + ab * 8 + (i0 & 7))) =
                *reinterpret_cast<const uint2*>(act);
        } else {
          *reinterpret_cast<uint2*>(p.out + (size_t)m * p.out_stride + n) =
              *reinterpret_cast<const uint2*>(o);
        }
      }
    }
  }
  if constexpr (EPI == 1) {
    // row maxima: across the lanes of a row (lane & 15), then across the K-part waves, then one atomicMax per row
    __shared__ float smax[KW][MT];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const float v = rows4_max(rowmax[mb]);
      if (lane < 16) smax[kw][mb * 16 + lane] = (m0 + mb * 16 + lane) < p.M ? v : 0.0f;
    }
    __syncthreads();
    if (threadIdx.x < MT) {
      float v = smax[0][threadIdx.x];
#pragma unroll
      for (int w = 1; w < KW; ++w) v = __builtin_fmaxf(v, smax[w][threadIdx.x]);
      const int m = m0 + threadIdx.x;
      if (m < p.M && !OMNI_GEMV_DBG_BIT(p, 4)) amax_raise(p.amax, m, blockIdx.x >> 3, v);
    }
  }
}

// Reduce SK int32 slabs and apply the epilogue.  One thread = 4 consecutive channels of one row.
template <int MODE>
__global__ __launch_bounds__(64) void splitk_epilogue_kernel(GemmArgs p, int sk) {
  const int n4 = p.N / 4;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)p.M * n4) return;
  const int m = idx / n4, n = (idx % n4) * 4;
  const int32_t* src = p.slab + (size_t)m * p.N + n;
  const size_t sstride = (size_t)p.M * p.N;
  v4i s0 = (v4i){0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
  int k = 0;
  for (; k + 4 <= sk; k += 4) {  // four independent 16-B loads in flight
    const v4i a0 = *reinterpret_cast<const v4i*>(src + (size_t)(k + 0) * sstride);
    const v4i a1 = *reinterpret_cast<const v4i*>(src + (size_t)(k + 1) * sstride);
    const v4i a2 = *reinterpret_cast<const v4i*>(src + (size_t)(k + 2) * sstride);
    const v4i a3 = *reinterpret_cast<const v4i*>(src + (size_t)(k + 3) * sstride);
    s0 += a0; s1 += a1; s2 += a2; s3 += a3;
  }
  for (; k < sk; ++k) s0 += *reinterpret_cast<const v4i*>(src + (size_t)k * sstride);
  const v4i s = (s0 + s1) + (s2 + s3);
  const float sa = (float)p.ascales[m];
  float as = 0.f;
  if constexpr (MODE == MODE_CHN) as = (float)p.asum[m];
  half_t o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float sw = (float)p.wscales[n + r];
    float sz = 0.f;
    if constexpr (MODE == MODE_CHN) sz = (float)p.wsz[n + r];
    o[r] = epilogue<MODE>(s[r], sw, sa, sz, as);
  }
  *reinterpret_cast<uint2*>(p.out + (size_t)m * p.out_stride + n) = *reinterpret_cast<const uint2*>(o);
}

// ---- host-side planning ----------------------------------------------------------------------
struct GemmPlan {
  int mb;      // 16-row blocks per workgroup tile
  int waves;   // waves (64-channel groups) per workgroup
  int sk;      // K splits (grid level, int32 slabs)
  int kslice;  // k per split
  int kw;      // K parts inside a workgroup (decode kernel)
  int mz;      // row tiles (grid.z) of the decode kernel: 2 for M = 65..128 (64-row tiles), up to 4 (32-row tiles)
  int narrow;  // 1: M = 33..128 on 32-row tiles (GemvCfg VAR = 1)
  int midm;    // 1: w4a8_midm_kernel (qgemm_midm.h): 128 channels x mb * 16 rows per workgroup, mz row tiles in grid.z
};

// Tuning hook (tests / bench sweeps): waves<=0 and sk<=0 restore the heuristic.
extern "C" void omni_gemm_set_plan_override(int waves, int sk);
// mode: -1 heuristic, 0 never the mid-M kernel, 1 wherever its shape conditions hold; sk > 0 forces its K split
extern "C" void omni_gemm_set_midm_override(int mode, int sk);
extern "C" void omni_gemm_get_plan(int M, int N, int K, int kalign, int* mb, int* waves, int* sk);
GemmPlan plan_gemm(int M, int N, int K, int kalign, bool deferred = false, bool w8 = false);

template <int MODE, int MB, int WAVES>
static void launch_variant(const GemmArgs& a, const GemmPlan& pl, hipStream_t st) {
  // prefill regime: chunked LDS staging, weights through L2, XCD-aware tile order
  GemmArgs b = a;
  b.tiles_n = (a.N / 64 + WAVES - 1) / WAVES;
  b.tiles_m = (a.M + MB * 16 - 1) / (MB * 16);
#if OMNI_GEMM_XCD_ORDER
  const int sbs = ((b.tiles_m + 7) / 8) * ((b.tiles_n + 7) / 8);
  dim3 grid(sbs * 64, 1, 1);
  if (b.tiles_m < 8) {      // (the super-block order needs whole blocks of 8 row tiles to spread over the XCDs)
    b.tile_linear = 1;
    grid.x = b.tiles_m * b.tiles_n;
  }
#else
  dim3 grid(b.tiles_n, 1, b.tiles_m);
  b.tiles_n = b.tiles_m = 0;
#endif
  // exact shapes (the models' projections at prefill): the straight-line kernel of qgemm_exact.h
  static const int exact_mode = omni_knob("OMNI_GEMM_EXACT", 1);   // 0: off (A/B, tuning builds)
  // (the exact kernel decodes the 1-D XCD tile order and stores 16 B per lane: rows of the output must start 16-B aligned)
  static_assert(OMNI_GEMM_XCD_ORDER == 1, "w4a8_gemm_exact_kernel reads tiles_m / tiles_n of the 1-D XCD-ordered grid");
  const bool out16 = (a.out_stride % 8) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
  if (MB == 8 && WAVES == 4 && exact_mode != 0 && a.M % 128 == 0 && a.N % 256 == 0 && a.K % KCHUNK == 0 && b.kslice >= a.K &&
      (size_t)a.M * a.K < ((size_t)1 << 32) && out16) {
    // (the form that stages activations through registers -- OMNI_GEMM_EXACT=2, A/B -- exists in tuning builds only: the release
    //  library launches the LDS-DMA form alone, and the register form's g128 / W8A8 instantiations spill)
#if defined(OMNI_TUNING) || !OMNI_GEMM_EXACT_DMA
    if (!(OMNI_GEMM_EXACT_DMA && exact_mode != 2)) {
      hipLaunchKernelGGL((w4a8_gemm_exact_kernel<MODE, false>), grid, dim3(256), 0, st, b);
      return;
    }
#endif
    hipLaunchKernelGGL((w4a8_gemm_exact_kernel<MODE, true>), grid, dim3(256), 0, st, b);
    return;
  }
  if (pl.sk > 1) {      // few tiles (decode at batch 129..512): K slices over grid.y -> int32 slabs -> the slab epilogue
    grid.y = pl.sk;
    if (MB == 8 && WAVES == 4 && exact_mode != 0 && OMNI_GEMM_EXACT_DMA && a.M % 128 == 0 && a.N % 256 == 0 && b.kslice % KCHUNK == 0 &&
        (size_t)a.M * a.K < ((size_t)1 << 32))
      hipLaunchKernelGGL((w4a8_gemm_exact_kernel<MODE, true, true>), grid, dim3(256), 0, st, b);
    else
      hipLaunchKernelGGL((w4a8_gemm_kernel<MB, MODE, WAVES, true, false>), grid, dim3(64 * WAVES), 0, st, b);
    const size_t total = (size_t)a.M * (a.N / 4);
    hipLaunchKernelGGL((splitk_epilogue_kernel<MODE>), dim3((total + 63) / 64), dim3(64), 0, st, a, pl.sk);
    return;
  }
  hipLaunchKernelGGL((w4a8_gemm_kernel<MB, MODE, WAVES, false, false>), grid, dim3(64 * WAVES), 0, st, b);
}

// Load policy of a decode-shape GEMV's weight stream, decided PER CALL: plain loads when the call's weight tensor is the
// one a preceding omni_prefetch_arm_gemm named (a row kernel pulled the head of its stream into the L2s -- one-shot entry,
// consumed here), non-temporal loads otherwise (streamed once from HBM).  qgemm_plan.hip.
bool take_prefetched_weight(const void* weight);

template <int MODE, int MB, bool TO_SLAB, bool NT>
static void launch_gemv_kernel_nt(const GemmArgs& a, const GemmPlan& pl, hipStream_t st) {
  constexpr int WAVES = GemvCfg<MB, MODE>::WAVES;
  dim3 grid((a.N / 64 + WAVES - 1) / WAVES, pl.sk, pl.mz);
  if constexpr (MB == 1) {
    if (pl.kw == 4) {
      hipLaunchKernelGGL((w4a8_gemv_kernel<1, MODE, TO_SLAB, 4, NT>), grid, dim3(256), 0, st, a);
      return;
    }
    if (pl.kw == 2) {
      hipLaunchKernelGGL((w4a8_gemv_kernel<1, MODE, TO_SLAB, 2, NT>), grid, dim3(128), 0, st, a);
      return;
    }
  }
  if constexpr (MB == 4 && GemvCfg<MB, MODE>::MAX_KW == 4) {
    if (pl.kw == 4 && pl.mz == 2) {
      hipLaunchKernelGGL((w4a8_gemv_kernel<MB, MODE, TO_SLAB, 4, NT, 2>), dim3(grid.x, grid.y, 1), dim3(512), 0, st, a);
      return;
    }
    if (pl.kw == 4) {
      hipLaunchKernelGGL((w4a8_gemv_kernel<MB, MODE, TO_SLAB, 4, NT>), grid, dim3(256), 0, st, a);
      return;
    }
  }
  if constexpr (MB == 4) {
    if (pl.kw == 2 && pl.mz == 2) {   // both 64-row tiles of a channel group in one workgroup (4 waves, 128 KiB of LDS)
      hipLaunchKernelGGL((w4a8_gemv_kernel<MB, MODE, TO_SLAB, 2, NT, 2>), dim3(grid.x, grid.y, 1), dim3(256), 0, st, a);
      return;
    }
  }
  if constexpr (MB == 2 && MODE != MODE_W8) {
    if (pl.narrow) {     // M = 33..128 as 32-row tiles (grid.z row tiles), four K parts per workgroup
      if (pl.kw == 4)
        hipLaunchKernelGGL((w4a8_gemv_kernel<MB, MODE, TO_SLAB, 4, NT, 1, 0, false, 1>), grid, dim3(256), 0, st, a);
      else if (pl.kw == 2)
        hipLaunchKernelGGL((w4a8_gemv_kernel<MB, MODE, TO_SLAB, 2, NT, 1, 0, false, 1>), grid, dim3(128), 0, st, a);
      else
        hipLaunchKernelGGL((w4a8_gemv_kernel<MB, MODE, TO_SLAB, 1, NT, 1, 0, false, 1>), grid, dim3(64), 0, st, a);
      return;
    }
  }
  if constexpr (MB == 2 || MB == 4) {
    if (pl.kw == 2) {
      hipLaunchKernelGGL((w4a8_gemv_kernel<MB, MODE, TO_SLAB, 2, NT>), grid, dim3(128), 0, st, a);
      return;
    }
  }
  hipLaunchKernelGGL((w4a8_gemv_kernel<MB, MODE, TO_SLAB, 1, NT>), grid, dim3(64 * WAVES), 0, st, a);
}

template <int MODE, int MB, bool TO_SLAB>
static void launch_gemv_kernel(const GemmArgs& a, const GemmPlan& pl, hipStream_t st) {
  if (take_prefetched_weight(a.W)) launch_gemv_kernel_nt<MODE, MB, TO_SLAB, false>(a, pl, st);
  else launch_gemv_kernel_nt<MODE, MB, TO_SLAB, true>(a, pl, st);
}

// mid-M kernel (qgemm_midm.h): grid = (128-channel tiles, K slices, row tiles)
template <int MODE, bool TO_SLAB>
static void launch_midm(const GemmArgs& a, const GemmPlan& pl, hipStream_t st) {
  const dim3 grid(a.N / 128, pl.sk, pl.mz), block(512);
  const bool nt = !take_prefetched_weight(a.W);
  GemmArgs b = a;
  b.tile_linear = (!TO_SLAB && (a.out_stride % 8) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0) ? 1 : 0;   // 16-B stores
  if (pl.mb == 8) {
    if (nt) hipLaunchKernelGGL((w4a8_midm_kernel<8, MODE, TO_SLAB, true>), grid, block, 0, st, b);
    else hipLaunchKernelGGL((w4a8_midm_kernel<8, MODE, TO_SLAB, false>), grid, block, 0, st, b);
  } else {
    if (nt) hipLaunchKernelGGL((w4a8_midm_kernel<4, MODE, TO_SLAB, true>), grid, block, 0, st, b);
    else hipLaunchKernelGGL((w4a8_midm_kernel<4, MODE, TO_SLAB, false>), grid, block, 0, st, b);
  }
}

template <int MODE, int MB>
static void launch_gemv(const GemmArgs& a, const GemmPlan& pl, hipStream_t st) {
  if (pl.sk > 1) {
    launch_gemv_kernel<MODE, MB, true>(a, pl, st);
    const size_t total = (size_t)a.M * (a.N / 4);
    hipLaunchKernelGGL((splitk_epilogue_kernel<MODE>), dim3((total + 63) / 64), dim3(64), 0, st, a, pl.sk);
  } else {
    launch_gemv_kernel<MODE, MB, false>(a, pl, st);
  }
}

template <int MODE>
static int launch_gemm(GemmArgs a, void* ws, size_t ws_bytes, hipStream_t st) {
  if (a.M < 1 || a.N % 64 != 0 || a.K % 64 != 0 || a.K < 64) return OMNI_EINVAL;
  if (MODE == MODE_GRP && a.K % 128 != 0) return OMNI_EINVAL;
  GemmPlan pl = plan_gemm(a.M, a.N, a.K, MODE == MODE_GRP ? 128 : 64, false, MODE == MODE_W8);
  if (pl.sk > 1) {
    const size_t need = (size_t)pl.sk * a.M * a.N * sizeof(int32_t);
    if (ws == nullptr || ws_bytes < need) return OMNI_ENOMEM;
    a.slab = static_cast<int32_t*>(ws);
  }
  a.kslice = pl.kslice;
  if (pl.midm) {
    if (pl.sk > 1) {
      launch_midm<MODE, true>(a, pl, st);
      const size_t total = (size_t)a.M * (a.N / 4);
      hipLaunchKernelGGL((splitk_epilogue_kernel<MODE>), dim3((total + 63) / 64), dim3(64), 0, st, a, pl.sk);
    } else {
      launch_midm<MODE, false>(a, pl, st);
    }
    return omni_launch_status();
  }
  if (a.M > 128) {
#ifdef OMNI_GEMM_TILE_M64
    launch_variant<MODE, 4, 4>(a, pl, st);
#else
    launch_variant<MODE, 8, 4>(a, pl, st);
#endif
  } else {
    switch (pl.mb) {
      case 1: launch_gemv<MODE, 1>(a, pl, st); break;
      case 2: launch_gemv<MODE, 2>(a, pl, st); break;
      default: launch_gemv<MODE, 4>(a, pl, st); break;
    }
  }
  return omni_launch_status();
}

// ---- row-kernel-free forms (fused extension) ---------------------------------------------------------------
template <int MODE, bool NT>
static int launch_gemv_silu_nt(const GemmArgs& a, const GemmPlan& pl, hipStream_t st) {
  dim3 grid(a.N / 64, 1, 1);
  switch (pl.kw) {
    case 4: hipLaunchKernelGGL((w4a8_gemv_kernel<1, MODE, false, 4, NT, 1, 1, false>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((w4a8_gemv_kernel<1, MODE, false, 2, NT, 1, 1, false>), grid, dim3(128), 0, st, a); break;
    default: hipLaunchKernelGGL((w4a8_gemv_kernel<1, MODE, false, 1, NT, 1, 1, false>), grid, dim3(64), 0, st, a); break;
  }
  return omni_launch_status();
}

// gate_up projection with silu_and_mul fused into the epilogue: out = act fp16 [M, N/2], a.amax raised per row.
// M <= 16 (the single-wave 16-row tile) and a plan without a grid-level K split (K <= 4096 per workgroup).
template <int MODE>
static int launch_gemm_silu(GemmArgs a, hipStream_t st) {
  if (a.M < 1 || a.M > 16 || a.N % 128 != 0 || a.K % 64 != 0 || a.K < 64 || !a.amax || !a.out) return OMNI_EINVAL;
  if (MODE == MODE_GRP && a.K % 128 != 0) return OMNI_EINVAL;
  GemmPlan pl = plan_gemm(a.M, a.N, a.K, MODE == MODE_GRP ? 128 : 64, false, MODE == MODE_W8);
  if (pl.sk != 1 || pl.mb != 1) return OMNI_EINVAL;
  a.kslice = pl.kslice;
  OMNI_GEMV_SET_DBG(a);
  return take_prefetched_weight(a.W) ? launch_gemv_silu_nt<MODE, false>(a, pl, st) : launch_gemv_silu_nt<MODE, true>(a, pl, st);
}

template <int MODE, bool NT>
static int launch_gemv_f16_nt(const GemmArgs& a, const GemmPlan& pl, hipStream_t st) {
  dim3 grid(a.N / 64, pl.sk + 1, 1);     // first grid row: the rider workgroups (one per activation row)
  switch (pl.kw) {
    case 4: hipLaunchKernelGGL((w4a8_gemv_kernel<1, MODE, true, 4, NT, 1, 0, true>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((w4a8_gemv_kernel<1, MODE, true, 2, NT, 1, 0, true>), grid, dim3(128), 0, st, a); break;
    default: return OMNI_EINVAL;
  }
  return omni_launch_status();
}

// o / down projection from fp16 activations + producer row maxima: int32 split-K slabs (like launch_gemm_partial) plus
// the sum / scale the slab consumer needs, written by rider workgroups.  M <= 16; K <= 16384 (the rider parks a row
// as f32 in the workgroup's staging LDS); the row must be whole 8-step rounds per wave.
template <int MODE>
static int launch_gemm_partial_f16(GemmArgs a, void* slab, size_t slab_bytes, int* sk_out, hipStream_t st) {
  if (a.M < 1 || a.M > 16 || a.N % 64 != 0 || a.K % 64 != 0 || a.K < 64 || !slab || !sk_out || !a.A16 || !a.amax ||
      !a.scale_out)
    return OMNI_EINVAL;
  if (MODE == MODE_GRP && a.K % 128 != 0) return OMNI_EINVAL;
  if (a.N / 64 < a.M) return OMNI_EINVAL;                                   // one rider per row in a grid row of N/64
  GemmPlan pl = plan_gemm(a.M, a.N, a.K, MODE == MODE_GRP ? 128 : 64, true, MODE == MODE_W8);
  if (pl.mb != 1 || pl.kw < 2) return OMNI_EINVAL;
  if (a.sum_out && (size_t)a.K * sizeof(float) > (size_t)pl.kw * 2 * 16 * GemvCfg<1, MODE>::AR * KSTEP) return OMNI_EINVAL;   // rider LDS (ordered row sum only)
  if (slab_bytes < (size_t)pl.sk * a.M * a.N * sizeof(int32_t)) return OMNI_ENOMEM;
  a.slab = static_cast<int32_t*>(slab);
  a.kslice = pl.kslice;
  *sk_out = pl.sk;
  OMNI_GEMV_SET_DBG(a);
  return take_prefetched_weight(a.W) ? launch_gemv_f16_nt<MODE, false>(a, pl, st) : launch_gemv_f16_nt<MODE, true>(a, pl, st);
}

// Deferred-epilogue variant (fused extension): only the int32 split-K slabs are produced; the consumer
// kernel (omni_splitk_add_rms_norm_general_fuse_sum) reduces them and applies the epilogue.
template <int MODE>
static int launch_gemm_partial(GemmArgs a, void* slab, size_t slab_bytes, int* sk_out, hipStream_t st) {
  if (a.M < 1 || a.M > 512 || a.N % 64 != 0 || a.K % 64 != 0 || a.K < 64 || !slab || !sk_out) return OMNI_EINVAL;
  if (MODE == MODE_GRP && a.K % 128 != 0) return OMNI_EINVAL;
  GemmPlan pl = plan_gemm(a.M, a.N, a.K, MODE == MODE_GRP ? 128 : 64, true, MODE == MODE_W8);
  if (slab_bytes < (size_t)pl.sk * a.M * a.N * sizeof(int32_t)) return OMNI_ENOMEM;
  a.slab = static_cast<int32_t*>(slab);
  a.kslice = pl.kslice;
  if (pl.midm) {
    launch_midm<MODE, true>(a, pl, st);
    *sk_out = pl.sk;
    return omni_launch_status();
  }
  if (a.M > 128) {      // decode batches of 129 .. 512 rows: the 128 x 256 tile with K slices over grid.y, slabs only
    GemmArgs b = a;
    b.tiles_n = (a.N / 64 + 3) / 4;
    b.tiles_m = (a.M + 127) / 128;
    b.tile_linear = 1;
    if (OMNI_GEMM_EXACT_DMA && a.M % 128 == 0 && a.N % 256 == 0 && b.kslice % KCHUNK == 0 && (size_t)a.M * a.K < ((size_t)1 << 32))
      hipLaunchKernelGGL((w4a8_gemm_exact_kernel<MODE, true, true>), dim3(b.tiles_m * b.tiles_n, pl.sk, 1), dim3(256), 0, st, b);
    else
      hipLaunchKernelGGL((w4a8_gemm_kernel<8, MODE, 4, true, false>), dim3(b.tiles_m * b.tiles_n, pl.sk, 1), dim3(256), 0, st, b);
    *sk_out = pl.sk;
    return omni_launch_status();
  }
  switch (pl.mb) {
    case 1: launch_gemv_kernel<MODE, 1, true>(a, pl, st); break;
    case 2: launch_gemv_kernel<MODE, 2, true>(a, pl, st); break;
    default: launch_gemv_kernel<MODE, 4, true>(a, pl, st); break;
  }
  *sk_out = pl.sk;
  return omni_launch_status();
}

}  // namespace omni

