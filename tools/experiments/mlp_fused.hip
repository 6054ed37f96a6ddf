// The MLP half of the W4A8 decode layer as ONE persistent launch (fused extension, nothing upstream):
//
//     residual += o_proj epilogue(split-K slabs)  ->  rms_norm_general_fuse_sum  ->  gate_up GEMV + silu_and_mul
//     ->  per-token quantisation  ->  down GEMV (int32 split-K slabs for the next slab consumer)
//
// i.e. llama_w4a8_unpad.py:425-436 at decode shape, which the launch-per-kernel path (fusion level 3) runs as three dependent
// launches (add + norm + quant row kernel 8.5 us, gate_up + SiLU GEMV 13.2 us, down GEMV 7.7 us at bs = 16): two kernel
// boundaries at which every CU drains its weight stream, and a row kernel during which 240 of 256 CUs only prefetch into L2.
// Here 256 workgroups (one per CU) stay resident for the whole sequence and the packed weights never stop streaming:
//
//   * every wave keeps a RING of 16 k-steps (32 KiB) of packed weights in flight in registers -- 128 KiB per CU, 32 MiB on the
//     chip -- and refills a step's registers for the NEXT unit of work right after unpacking them, across the boundaries
//     between gate_up units, and (re-armed behind each hand-off's arrival) across the hand-offs themselves: while 16 "service"
//     workgroups run the norm rows, and while the chip waits for the last gate_up unit, the next weights are already landing;
//   * hand-off 1 (norm -> gate_up): the service workgroups publish the int8 codes in MFMA OPERAND ORDER ([k-step][lane][16 B],
//     written through with agent-scope stores) plus {scale, sum} per row and arrive on one counter; every other workgroup
//     polls that word with one lane, takes ONE agent-scope acquire and loads its B operands (64 VGPRs, kept for both of its
//     gate_up units) with plain loads;
//   * hand-off 2 (gate_up -> down): the fp16 activation silu(gate) * up goes out in the operand order of down_proj's K
//     dimension, row maxima are raised with device-scope integer atomicMax on 8 x 16 sharded words (exact, order independent),
//     arrivals are counted on 8 sharded counters (32 arrivals each); the down units quantise their activation slice on the
//     fly (code = rni_sat(x * (127 / amax)): invoke_quant's arithmetic) exactly like omni_w4a8_per_chn_gemm_partial_f16;
//   * the ordered row sum / scale the slab consumer needs (invoke_quant_fuse_sum's reduction tree) is replayed by the
//     service workgroups after hand-off 2 ("rider", as in qgemm_kernel.h).
//
// Every spin is bounded (a workgroup that is not resident sets the error word instead of hanging the queue); counters are
// zeroed once per decode step by the caller and compared against (phase + 1) * arrivals, phase = the layer's index, a
// launch-time constant of each captured launch.  All arithmetic is the level-3 kernels' (same epilogue<>, silu_mul_h,
// quant4_f16, general_norm_v2_row, ordered row sum): results are bit-identical to the three-launch sequence
// (tests/test_mlp_fused_gpu.py, tests/test_runtime_gpu.py).
#include "qgemm_kernel.h"
#include "row_kernels.h"

namespace omni {

constexpr int MLP_WGS = 256;           // one workgroup per CU; all must be resident (they wait on each other)
constexpr int MLP_THREADS = 512;
constexpr int MLP_WAVES = MLP_THREADS / 64;
constexpr int MLP_RING = 8;            // k-steps of packed weights in flight per wave
constexpr int MLP_SERVICE = 16;        // workgroups 0..15 run the norm row / rider of activation row blockIdx.x
constexpr int MLP_SHARDS = 8;          // arrival counters / row-maximum words of hand-off 2
// counter words (uint32, zeroed once per decode step): one 64-B line each
constexpr int MLP_W_CNT1 = 0, MLP_W_CNT2 = 16, MLP_W_ERR = 16 + 16 * MLP_SHARDS, MLP_CNT_WORDS = MLP_W_ERR + 16;
constexpr int MLP_AMAX_WORDS = MLP_SHARDS * 16;       // per phase (layer): [shard][row]
constexpr int MLP_CLK_MARKS = 8;

struct MlpArgs {
  half_t* res;                 // [M, H] residual stream, updated in place
  const int32_t* o_slab;       // [sk_o][M][H] o_proj split-K slabs
  int sk_o;
  const half_t* o_ws;          // [H] o_proj weight scales / zero terms
  const half_t* o_wsz;
  const half_t* o_as;          // [M] scale / sum of o_proj's int8 input
  const half_t* o_asum;
  const half_t* gamma;         // [H] post-attention norm weight
  float eps;
  const uint8_t* Wgu;          // packed [2I, H/2]
  const half_t* gu_ws;         // [2I]
  const half_t* gu_wsz;
  const uint8_t* Wdn;          // packed [H, I/2]
  int32_t* dn_slab;            // [I/2048][M][H] down_proj split-K slabs (out)
  half_t* act_sum;             // [M] ordered row sum of the MLP activation (out)
  half_t* act_scale;           // [M] h(amax / 127) (out)
  uint32_t* cnt;               // [MLP_CNT_WORDS] zeroed once per step
  uint32_t* amax;              // [MLP_AMAX_WORDS] of THIS phase, zeroed once per step
  uint8_t* xq;                 // scratch [H/64][64][16 B] int8 codes in operand order
  uint32_t* sbmb;              // scratch [16] {scale, sum} fp16 pairs of the normed rows
  half_t* act;                 // scratch [I/64][64][16] fp16 activation in down_proj's operand order
  unsigned long long* clk;     // optional [MLP_WGS][MLP_CLK_MARKS] 100 MHz wall-clock marks
  int M, H, I, phase;
};

#define MLP_CLK(k)                                                                                   \
  do {                                                                                               \
    if (a.clk && threadIdx.x == 0) a.clk[(size_t)blockIdx.x * MLP_CLK_MARKS + (k)] = wall_clock64(); \
  } while (0)

__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one lane: poll until *w >= target (wrap-safe); bounded -- a lost workgroup raises the error word instead of hanging
__device__ __forceinline__ void wait_ge(const uint32_t* w, uint32_t target, uint32_t* err) {
  for (int spin = 0; spin < (1 << 22); ++spin) {
    if ((int32_t)(ld_agent(w) - target) >= 0) return;
    __builtin_amdgcn_s_sleep(2);
  }
  __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent64(void* p, uint64_t v) {      // written through to memory (sc1)
  __hip_atomic_store(reinterpret_cast<uint64_t*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// norm service sink: int8 codes of the row parked in LDS (natural order), scale / sum kept for the publish
struct SinkLds {
  uint8_t* codes_lds;
  uint32_t* pair_lds;       // [2]: scale bits, sum bits
  __device__ __forceinline__ void codes(int i, uint2 c) const { *reinterpret_cast<uint2*>(codes_lds + i) = c; }
  __device__ __forceinline__ void scale(half_t s) const { pair_lds[0] = __builtin_bit_cast(uint16_t, s); }
  __device__ __forceinline__ void sum(half_t s) const { pair_lds[1] = __builtin_bit_cast(uint16_t, s); }
};

#ifndef MLP_ACQUIRE_FENCE
#define MLP_ACQUIRE_FENCE 0      // 1: an agent-scope acquire (buffer_inv sc1, ~1.3 us) behind each hand-off's poll.  Not needed here:
#endif                           // every hand-off buffer is first touched by its consumer AFTER the hand-off (no line of it can sit in
                                 // the CU's L1 from earlier in the launch), its producers store write-through, polls bypass the L1

// 512 threads = 8 waves = two per SIMD (each other's LDS / MFMA latencies overlap; one wave per SIMD measured 4.5-5 us per
// 16-step round, profiles/r04_a): wave w owns the K eighth [512 w, 512 w + 512) of a gate_up unit (8 k-steps) and the K eighth
// [256 w, +256) of each down unit (4 + 4 k-steps).
// Two buffer levels per wave, 16 KiB each (256 KiB per CU, 64 MiB on the chip -- a round that REFILLS from memory is paced by
// the CU's miss path, 5 us per 128 KiB, so every byte that can be on the chip before a hand-off resolves is time saved):
//   * a LANDING SLOT in LDS, filled by LDS-DMA (global_load_lds: no registers, hidden from hipcc -- the wave counts its own
//     vmcnt: the DMAs are waited for with vmcnt(0) at points where everything older has to be complete anyway);
//   * the register ring (64 VGPRs) the MFMAs are fed from, refilled from the landing slot (ds_read) step by step.
// Stream of a heavy wave: unit A -> slot -> registers at once; unit B -> slot (lands during the norm / hand-off 1); round A
// consumes the registers and pulls B out of the slot, re-arming the slot with the down units' steps; round B consumes B;
// the down units are consumed straight from the slot after hand-off 2.
__global__ __launch_bounds__(MLP_THREADS, 2) void mlp_fused_kernel(MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // [0, 16 KiB)        partial sums of a GEMV epilogue, combined in two stages (waves 4..7 -> waves 0..3 -> row-block owners)
  // [16 KiB, +128 KiB) the 8 landing slots; before the first DMA (service workgroups) the norm row's f32 copy + int8 codes live
  //                    here, after the last down unit the rider's f32 copy of an activation row
  // then `red`, the {scale, sum} pair and the row maxima of the waves
  v4i* part = reinterpret_cast<v4i*>(smem);
  uint8_t* slots = smem + 16384;
  float* xs = reinterpret_cast<float*>(slots);
  uint8_t* codes_lds = reinterpret_cast<uint8_t*>(xs + a.H);
  float* red = reinterpret_cast<float*>(slots + 131072);
  uint32_t* pair_lds = reinterpret_cast<uint32_t*>(red + 96);
  float* smax = reinterpret_cast<float*>(pair_lds + 4);      // [8 waves][16 rows]

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave = K part
  const int G = MLP_WGS;
  const int U = a.I / 32;                                     // gate_up units (32 gate + 32 up channels, all of K) = down units
  const bool heavy = (G - 1 - b) < (U - G);                   // two gate_up and two down units (light: one of each)
  const uint32_t epoch = (uint32_t)a.phase + 1u;
  const bool service = b < MLP_SERVICE;
  MLP_CLK(0);

  // ---- weight addressing (qgemm_kernel.h: lane -> tile row lx, chunk (n3 = lc, k6 = le)) -------------------------------
  const int lx = (lane >> 3) & 1, lc = lane & 7, le = lane >> 4;
  const size_t chunk = (size_t)(lc * 4 + le) * 16;
  auto gu_base = [&](int u) -> const uint8_t* {      // this wave's 512-k eighth of gate tile row u / up tile row U + u
    const size_t trow = lx ? (size_t)U + u : (size_t)u;
    return a.Wgu + trow * (size_t)(a.H / 32) * 512 + (size_t)(w * (a.H / MLP_WAVES) / 32) * 512 + chunk;
  };
  auto dn_base = [&](int v) -> const uint8_t* {      // this wave's 256-k eighth of K part v / 64 of channel group v % 64
    const int g = v & 63, pt = v >> 6;
    const size_t trow = (size_t)(2 * g + lx);
    return a.Wdn + trow * (size_t)(a.I / 32) * 512 + (size_t)((pt * 2048 + w * (2048 / MLP_WAVES)) / 32) * 512 + chunk;
  };
  const int uA = b, uB = G + (G - 1 - b);            // gate_up units of this workgroup (uB: heavy only)
  const int vA = b, vB = G + (G - 1 - b);            // down units
  const uint8_t* dnA = dn_base(vA);
  const uint8_t* dnB = dn_base(heavy ? vB : vA);     // (light: the second half of the slot re-reads unit A, unused)

  constexpr int HR = MLP_RING / 2;
  uint8_t* slot = slots + (size_t)w * 16384;         // this wave's landing slot: [k-step][tile 0 | 1][64 lanes x 16 B]
  const uint8_t* slot_l = slot + (size_t)lane * 16;
  // steps [s0, s1) of the slot <- 2 KiB per step from p0 (steps 0..3) / p1 (steps 4..7), nobody but this wave reads the slot
  auto dma_steps = [&](const uint8_t* p0, const uint8_t* p1, int s0, int s1) {
#pragma unroll
    for (int s = 0; s < MLP_RING; ++s) {
      if (s < s0 || s >= s1) continue;
      const uint8_t* p = s < HR ? p0 + (size_t)s * 1024 : p1 + (size_t)(s - HR) * 1024;
      lds_dma16_untracked(p, slot + (size_t)(2 * s) * 1024);
      lds_dma16_untracked(p + 512, slot + (size_t)(2 * s + 1) * 1024);
    }
  };
  auto slot_step = [&](int s, uint4 (&wv)[2]) {
    wv[0] = *reinterpret_cast<const uint4*>(slot_l + (size_t)(2 * s) * 1024);
    wv[1] = *reinterpret_cast<const uint4*>(slot_l + (size_t)(2 * s + 1) * 1024);
  };
  uint4 wq[MLP_RING][2];
  auto unpack = [&](const uint4 (&wv)[2], v4i (&wa)[4]) {
    const uint4 t0 = wv[0], t1 = wv[1];
    const uint32_t d[2][4] = {{t0.x, t0.z, t1.x, t1.z}, {t0.y, t0.w, t1.y, t1.w}};
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        uint32_t u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q] = (d[y][q] >> (4 * x)) & 0x0F0F0F0Fu;
        wa[x * 2 + y] = (v4i){(int)u[0], (int)u[1], (int)u[2], (int)u[3]};
      }
  };
  // unit A -> slot -> registers; then the slot is re-armed with the next unit (heavy: gate_up unit B; light: the down units)
  auto start_stream = [&]() {
    dma_steps(gu_base(uA), gu_base(uA) + HR * 1024, 0, MLP_RING);
    drain_vmem();
#pragma unroll
    for (int s = 0; s < MLP_RING; ++s) slot_step(s, wq[s]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (the slot has been read: the DMAs below may overwrite it)
    if (heavy) dma_steps(gu_base(uB), gu_base(uB) + HR * 1024, 0, MLP_RING);
    else dma_steps(dnA, dnB, 0, MLP_RING);
  };

  // epilogue operands of the gate_up units (waves 0..3 finish row block ab = w of each unit): requested now
  const int i0 = (lane >> 4) * 4;
  const int fab = w & 3;
  auto gu_chan = [&](int u) -> int { return (i0 >> 3) * a.I + u * 32 + fab * 8 + (i0 & 7); };       // gate | up channel
  uint2 swA = *reinterpret_cast<const uint2*>(a.gu_ws + gu_chan(uA)), szA = *reinterpret_cast<const uint2*>(a.gu_wsz + gu_chan(uA));
  uint2 swB = swA, szB = szA;
  if (heavy) { swB = *reinterpret_cast<const uint2*>(a.gu_ws + gu_chan(uB)); szB = *reinterpret_cast<const uint2*>(a.gu_wsz + gu_chan(uB)); }

  // ---- P1: norm rows (service workgroups first, THEN their stream: their row loads must not queue behind the weights, and
  //      the row lives where the slots are); everybody else starts streaming at once ------------------------------------------
  if (!service) start_stream();
  if (service) {
    const int r = b;
    if (r < a.M) {
      SrcSlabAddChn src{a.res, a.o_slab, (size_t)a.M * a.H, a.sk_o, a.H, a.o_ws, a.o_wsz, a.o_as, a.o_asum, (half_t)0.0f, (half_t)0.0f};
      const SrcSlabAddChn row = src.at_row(r);
      SinkLds sink{codes_lds, pair_lds};
      general_norm_v2_row<MLP_THREADS, 4, true, SrcSlabAddChn, SinkLds>(row, a.gamma, sink, a.eps, a.H, 1024, xs, red);
    } else {
      for (int i = tid; i < a.H / 4; i += MLP_THREADS) reinterpret_cast<uint32_t*>(codes_lds)[i] = 0u;
      if (tid == 0) { pair_lds[0] = 0u; pair_lds[1] = 0u; }
    }
    __syncthreads();
    // publish the row in operand order: piece (k-step s, slot e) = dwords j = 0..3 of natural k = 64 s + 16 j + 4 e
    const uint32_t* c32 = reinterpret_cast<const uint32_t*>(codes_lds);
    for (int t = tid; t < a.H / 16; t += MLP_THREADS) {
      const int s = t >> 2, e = t & 3;
      const uint32_t d0 = c32[16 * s + e], d1 = c32[16 * s + 4 + e], d2 = c32[16 * s + 8 + e], d3 = c32[16 * s + 12 + e];
      uint8_t* dst = a.xq + ((size_t)s * 64 + (r + 16 * e)) * 16;
      st_agent64(dst, (uint64_t)d0 | ((uint64_t)d1 << 32));
      st_agent64(dst + 8, (uint64_t)d2 | ((uint64_t)d3 << 32));
    }
    if (tid == 0)
      __hip_atomic_store(a.sbmb + r, pair_lds[0] | (pair_lds[1] << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    drain_vmem();                                   // every storing wave: its write-through stores have landed
    __syncthreads();                                // (and every wave is done with the row in LDS: the slots may be written)
    if (tid == 0) __hip_atomic_fetch_add(a.cnt + MLP_W_CNT1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    MLP_CLK(1);
    start_stream();
  }

  // ---- hand-off 1: all 16 rows published ----------------------------------------------------------------------------
  if (tid == 0) wait_ge(a.cnt + MLP_W_CNT1, MLP_SERVICE * epoch, a.cnt + MLP_W_ERR);
#if MLP_ACQUIRE_FENCE
  if (w == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  __syncthreads();
  MLP_CLK(2);
  v4i breg[MLP_RING];                               // B operands of this wave's K eighth (both gate_up units)
#pragma unroll
  for (int s = 0; s < MLP_RING; ++s)
    breg[s] = *reinterpret_cast<const v4i*>(a.xq + ((size_t)(w * MLP_RING + s) * 64 + lane) * 16);
  uint32_t sbw = a.sbmb[lane & 15];
  // Everything the epilogues read from memory must be IN REGISTERS before the rounds start: hipcc otherwise sinks these
  // small loads to their first use, and a wait for them is then a wait for everything issued before (loads return in
  // order): measured, every round ended on a 5-us drain of its own prefetch (profiles/r04_a).
  asm volatile("" : "+v"(sbw), "+v"(swA.x), "+v"(swA.y), "+v"(szA.x), "+v"(szA.y), "+v"(swB.x), "+v"(swB.y), "+v"(szB.x), "+v"(szB.y));
  const float sa = (float)__builtin_bit_cast(half_t, (uint16_t)(sbw & 0xFFFFu));
  const float as = (float)__builtin_bit_cast(half_t, (uint16_t)(sbw >> 16));
#pragma unroll
  for (int s = 0; s < MLP_RING; ++s) asm volatile("" : "+v"(breg[s]));
  drain_vmem();                                     // (everything issued so far has landed -- the slot's DMAs included)

  // ---- P2: gate_up units ---------------------------------------------------------------------------------------------
  float rowmax = 0.0f;                              // max |act| of row (lane & 15) over the channels this lane finished
  // one round = the 8 ring steps of one gate_up unit.  PULL: the registers of a step are refilled from the landing slot
  // (gate_up unit B) right after they were unpacked, and each half of the slot is re-armed with the down units' steps once it
  // has been read.
  auto gu_round = [&](auto pull_tag, v4i (&acc)[4]) {
    constexpr bool PULL = decltype(pull_tag)::value;
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) acc[ab] = (v4i){0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < MLP_RING; ++s) {
      v4i wa[4];
      unpack(wq[s], wa);
      if constexpr (PULL) slot_step(s, wq[s]);
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) acc[ab] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[ab], breg[s], acc[ab], 0, 0, 0);
      if constexpr (PULL) {
        if (s == HR - 1 || s == MLP_RING - 1) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this half of the slot is in registers
          dma_steps(dnA, dnB, s == HR - 1 ? 0 : HR, s == HR - 1 ? HR : MLP_RING);
        }
      }
    }
  };
  // K parts meet in LDS in two stages (16 KiB): waves 4..7 -> waves 0..3, then waves 0..3 -> the owner of each row block
  // (wave ab); the owners run the epilogue, silu_and_mul, and store the fp16 activation in operand order
  auto combine = [&](v4i (&acc)[4]) -> v4i {
    if (w >= 4) {
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) part[((w - 4) * 4 + ab) * 64 + lane] = acc[ab];
    }
    __syncthreads();
    if (w < 4) {
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) acc[ab] += part[(w * 4 + ab) * 64 + lane];
    }
    __syncthreads();
    if (w < 4) {
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) part[(w * 4 + ab) * 64 + lane] = acc[ab];
    }
    __syncthreads();
    v4i a4 = (v4i){0, 0, 0, 0};
    if (w < 4) {
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) a4 += part[(ww * 4 + w) * 64 + lane];
    }
    __syncthreads();                                // (the next unit's partials may overwrite)
    return a4;
  };
  auto gu_finish = [&](v4i (&acc)[4], int u, uint2 swv, uint2 szv) {
    const v4i a4 = combine(acc);
    if (w < 4) {
      typedef _Float16 v4h_t __attribute__((ext_vector_type(4)));
      const v4h_t sw4 = __builtin_bit_cast(v4h_t, swv), sz4 = __builtin_bit_cast(v4h_t, szv);
      half_t o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = epilogue<MODE_CHN>(a4[r], (float)sw4[r], sa, (float)sz4[r], as);
      // lanes < 32 hold the fp16 gate outputs, lanes >= 32 the up outputs of the same (row, 4 channels)
      const uint2 mine = *reinterpret_cast<const uint2*>(o);
      const uint2 other = make_uint2((uint32_t)__shfl_xor((int)mine.x, 32, 64), (uint32_t)__shfl_xor((int)mine.y, 32, 64));
      const uint2 g2 = lane < 32 ? mine : other, u2 = lane < 32 ? other : mine;
      const v4h_t g4 = __builtin_bit_cast(v4h_t, g2), u4 = __builtin_bit_cast(v4h_t, u2);
      half_t act[4];
      float mx = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        act[r] = silu_mul_h(g4[r], u4[r]);
        mx = __builtin_fmaxf(mx, __builtin_fabsf((float)act[r]));
      }
      rowmax = __builtin_fmaxf(rowmax, mx);
      if (lane < 32) {
        // activation channels n .. n + 3 of row m: k-step n / 64 of down_proj, piece j = (n % 64) / 16, slot e = (n % 16) / 4
        const int n = u * 32 + w * 8 + (i0 & 7), m = lane & 15;
        const int s = n >> 6, j = (n >> 4) & 3, e = (n >> 2) & 3;
        st_agent64(a.act + ((size_t)s * 64 + (m + 16 * e)) * 16 + 4 * j, *reinterpret_cast<const uint64_t*>(act));
      }
    }
  };
  {
    v4i acc[4];
    if (heavy) {
      gu_round(BoolTag<true>{}, acc);
      gu_finish(acc, uA, swA, szA);
      MLP_CLK(3);
      gu_round(BoolTag<false>{}, acc);
      gu_finish(acc, uB, swB, szB);
    } else {
      gu_round(BoolTag<false>{}, acc);
      gu_finish(acc, uA, swA, szA);
      MLP_CLK(3);
    }
  }
  // ---- hand-off 2: arrive (row maxima raised, activation written through), wait -----------------------------------------
  {
    const float v = rows4_max(rowmax);
    if (lane < 16) smax[w * 16 + lane] = v;         // (waves 4..7 never finished a channel: zeros)
    __syncthreads();
    if (tid < 16) {
      float mxr = smax[tid];
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) mxr = __builtin_fmaxf(mxr, smax[ww * 16 + tid]);
      __hip_atomic_fetch_max(a.amax + (b & (MLP_SHARDS - 1)) * 16 + tid, __builtin_bit_cast(uint32_t, mxr), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
    }
    drain_vmem();                                   // (the activation stores -- and the down units' DMAs, issued a round ago)
    __syncthreads();
    if (tid == 0)
      __hip_atomic_fetch_add(a.cnt + MLP_W_CNT2 + 16 * (b & (MLP_SHARDS - 1)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  MLP_CLK(4);
  if (tid < MLP_SHARDS) wait_ge(a.cnt + MLP_W_CNT2 + 16 * tid, (uint32_t)(G / MLP_SHARDS) * epoch, a.cnt + MLP_W_ERR);
#if MLP_ACQUIRE_FENCE
  if (w == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  __syncthreads();
  MLP_CLK(5);
  // row maxima -> 127 / amax of row (lane & 15)
  float amax_row;
  {
    uint32_t mbits = 0;
#pragma unroll
    for (int sh = 0; sh < MLP_SHARDS; ++sh) mbits = max(mbits, a.amax[sh * 16 + (lane & 15)]);
    amax_row = __builtin_bit_cast(float, mbits);
  }
  const float qlane = quant_multiplier(amax_row);

  // ---- P3: down units (fp16 activation slice quantised on the fly -> int32 split-K slab of K part v / 64) ---------------
  // B operands: both units' slices of this wave (4 + 4 k-steps) requested together, quantised into 32 registers; the packed
  // weights are consumed straight from the landing slot
  v4i bq[MLP_RING];
  {
    uint4 raw[MLP_RING][2];
#pragma unroll
    for (int s = 0; s < MLP_RING; ++s) {
      const int v = (s < HR || !heavy) ? vA : vB;
      const int ks = ((v >> 6) * 2048 + w * (2048 / MLP_WAVES)) / 64 + (s & (HR - 1));
      const half_t* src = a.act + ((size_t)ks * 64 + lane) * 16;
      raw[s][0] = *reinterpret_cast<const uint4*>(src);
      raw[s][1] = *reinterpret_cast<const uint4*>(src + 8);
    }
#pragma unroll
    for (int s = 0; s < MLP_RING; ++s)
      bq[s] = (v4i){(int)quant4_f16(raw[s][0].x, raw[s][0].y, qlane), (int)quant4_f16(raw[s][0].z, raw[s][0].w, qlane),
                    (int)quant4_f16(raw[s][1].x, raw[s][1].y, qlane), (int)quant4_f16(raw[s][1].z, raw[s][1].w, qlane)};
  }
  auto dn_unit = [&](int v, auto half_tag) {
    constexpr int S0 = decltype(half_tag)::value ? HR : 0;       // slot steps S0 .. S0 + 3
    const int g = v & 63, pt = v >> 6;
    v4i acc[4];
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) acc[ab] = (v4i){0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < HR; ++s) {
      uint4 wv[2];
      slot_step(S0 + s, wv);
      v4i wa[4];
      unpack(wv, wa);
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) acc[ab] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[ab], bq[S0 + s], acc[ab], 0, 0, 0);
    }
    const v4i a4 = combine(acc);
    if (w < 4) {
      const int m = lane & 15;
      if (m < a.M) {
        const int n = g * 64 + (i0 >> 3) * 32 + w * 8 + (i0 & 7);
        *reinterpret_cast<v4i*>(a.dn_slab + ((size_t)pt * a.M + m) * a.H + n) = a4;
      }
    }
  };
  dn_unit(vA, BoolTag<false>{});
  if (heavy) dn_unit(vB, BoolTag<true>{});
  MLP_CLK(6);

  // ---- rider: invoke_quant_fuse_sum's ordered row sum + scale of activation row b (service workgroups) ------------------
  if (service && b < a.M) {
    const int r = b;
    const float rmax = __shfl(amax_row, r, 64);
    __syncthreads();                                 // (every wave is done with its slot: the row goes where the slots are)
    for (int i = tid * VT; i < a.I; i += MLP_THREADS * VT) {
      // 8 consecutive channels i .. i + 7 = pieces (s, j, e) and (s, j, e + 1) of the operand-order activation
      const int s = i >> 6, j = (i >> 4) & 3, e = (i >> 2) & 3;
      typedef _Float16 v4h_t __attribute__((ext_vector_type(4)));
      const v4h_t lo = *reinterpret_cast<const v4h_t*>(a.act + ((size_t)s * 64 + (r + 16 * e)) * 16 + 4 * j);
      const v4h_t hi = *reinterpret_cast<const v4h_t*>(a.act + ((size_t)s * 64 + (r + 16 * (e + 1))) * 16 + 4 * j);
      *reinterpret_cast<v4f*>(xs + i) = (v4f){(float)lo[0], (float)lo[1], (float)lo[2], (float)lo[3]};
      *reinterpret_cast<v4f*>(xs + i + 4) = (v4f){(float)hi[0], (float)hi[1], (float)hi[2], (float)hi[3]};
    }
    __syncthreads();
    float sv[1][VT], tot[1];
    ordered_partials<1>(xs, tid, 1024, a.I, sv, [](float (&v)[1][VT], int e, float val) { v[0][e] = v[0][e] + val; });
    tree_sum8<1>(sv, red, tid, 32, tot);
    if (tid == 0) {
      a.act_sum[r] = (half_t)tot[0];
      a.act_scale[r] = (half_t)(rmax / 127.0f);
    }
  }
  MLP_CLK(7);
}

}  // namespace omni

using namespace omni;

extern "C" size_t omni_mlp_fused_counter_words(int layers) {
  return layers < 1 ? 0 : (size_t)MLP_CNT_WORDS + (size_t)layers * MLP_AMAX_WORDS;
}
extern "C" size_t omni_mlp_fused_scratch_bytes(int hidden, int inter) {
  if (hidden < 64 || inter < 64) return 0;
  // xq (hidden bytes x 16 rows) | {scale, sum} words | fp16 activation (16 rows) | phase clocks
  return (size_t)hidden * 16 + 256 + (size_t)inter * 16 * 2 + (size_t)MLP_WGS * MLP_CLK_MARKS * 8;
}

// 1 when the persistent MLP launch takes this layer on this device: M <= 16 rows, hidden = 4096 (a wave's K quarter = one
// ring of 16 k-steps), intermediate size a multiple of 2048 with 256 <= inter / 32 <= 512 units (one or two per workgroup),
// and a device whose 256 CUs can hold the 256 workgroups together.
extern "C" int omni_mlp_fused_ok(int M, int hidden, int inter) {
  if (M < 1 || M > 16 || hidden != 4096 || inter % 2048 != 0 || inter / 32 < MLP_WGS || inter / 32 > 2 * MLP_WGS) return 0;
  if ((size_t)inter * 4 > 131072) return 0;                                      // a row as f32 in the landing slots' LDS
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  return prop.multiProcessorCount >= MLP_WGS ? 1 : 0;
}

// residual += o_proj epilogue(slabs); norm + quant; gate_up + silu_and_mul; quant; down_proj -> split-K slabs, in one launch.
// `counters`: omni_mlp_fused_counter_words(layers) uint32 words ZEROED ONCE PER DECODE STEP by the caller (before the first
// layer's launch); `phase`: the layer's index (launch l of a step compares its arrivals against (l + 1) x the per-launch
// count); `scratch`: omni_mlp_fused_scratch_bytes() bytes, contents dead between launches; `clocks != 0`: the launch records
// its phase marks in the scratch tail (tools/mlp_timeline.py).  Only enqueues.
extern "C" int omni_w4a8_per_chn_mlp_fused(void* residual_f16, const void* o_slab_i32, int sk_o, const void* o_wscales,
                                           const void* o_wsz, const void* o_ascales, const void* o_asum, const void* gamma_f16,
                                           float eps, const void* gu_qweight, const void* gu_wscales, const void* gu_wsz,
                                           const void* dn_qweight, void* dn_slab_i32, size_t dn_slab_bytes, int* sk_out,
                                           void* act_sum_f16, void* act_scale_f16, void* counters, int layers, int phase,
                                           void* scratch, size_t scratch_bytes, int clocks, int M, int hidden, int inter,
                                           void* stream) {
  if (!residual_f16 || !o_slab_i32 || !o_wscales || !o_wsz || !o_ascales || !o_asum || !gamma_f16 || !gu_qweight ||
      !gu_wscales || !gu_wsz || !dn_qweight || !dn_slab_i32 || !sk_out || !act_sum_f16 || !act_scale_f16 || !counters || !scratch)
    return OMNI_EINVAL;
  if (sk_o < 1 || sk_o > 8 || phase < 0 || phase >= layers || omni_mlp_fused_ok(M, hidden, inter) != 1) return OMNI_EINVAL;
  const int sk = inter / 2048;
  if (dn_slab_bytes < (size_t)sk * M * hidden * sizeof(int32_t)) return OMNI_ENOMEM;
  if (scratch_bytes < omni_mlp_fused_scratch_bytes(hidden, inter)) return OMNI_ENOMEM;
  MlpArgs a{};
  a.res = (half_t*)residual_f16; a.o_slab = (const int32_t*)o_slab_i32; a.sk_o = sk_o;
  a.o_ws = (const half_t*)o_wscales; a.o_wsz = (const half_t*)o_wsz; a.o_as = (const half_t*)o_ascales; a.o_asum = (const half_t*)o_asum;
  a.gamma = (const half_t*)gamma_f16; a.eps = eps;
  a.Wgu = (const uint8_t*)gu_qweight; a.gu_ws = (const half_t*)gu_wscales; a.gu_wsz = (const half_t*)gu_wsz;
  a.Wdn = (const uint8_t*)dn_qweight; a.dn_slab = (int32_t*)dn_slab_i32;
  a.act_sum = (half_t*)act_sum_f16; a.act_scale = (half_t*)act_scale_f16;
  a.cnt = (uint32_t*)counters;
  a.amax = (uint32_t*)counters + MLP_CNT_WORDS + (size_t)phase * MLP_AMAX_WORDS;
  uint8_t* sc = (uint8_t*)scratch;
  a.xq = sc; a.sbmb = (uint32_t*)(sc + (size_t)hidden * 16); a.act = (half_t*)(sc + (size_t)hidden * 16 + 256);
  a.clk = clocks ? (unsigned long long*)(sc + (size_t)hidden * 16 + 256 + (size_t)inter * 16 * 2) : nullptr;
  a.M = M; a.H = hidden; a.I = inter; a.phase = phase;
  *sk_out = sk;
  // > 80 KiB of LDS per workgroup: one workgroup per CU whatever the register allocation (they wait on each other)
  // 16 KiB of partial sums + eight 16-KiB landing slots (a row as f32 lives there before / after the stream) + reduction
  // scratch: one workgroup per CU (they wait on each other)
  const size_t lds = 16384 + 131072 + 96 * 4 + 16 + 8 * 16 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)mlp_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      (void)hipGetLastError();
    attr_set = true;
  }
  hipLaunchKernelGGL(mlp_fused_kernel, dim3(MLP_WGS), dim3(MLP_THREADS), lds, (hipStream_t)stream, a);
  return omni_launch_status();
}

// error word of the counters (a bounded spin gave up: a workgroup was not resident); reading it synchronises the stream
extern "C" int omni_mlp_fused_error(const void* counters, void* stream) {
  if (!counters) return OMNI_EINVAL;
  uint32_t v = 0;
  if (hipMemcpyAsync(&v, (const uint32_t*)counters + MLP_W_ERR, 4, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
    return OMNI_ELAUNCH;
  return v ? 1 : 0;
}
