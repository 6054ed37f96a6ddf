// (norm -> GEMV) pairs of the decode layer as ONE launch with heterogeneous workgroups (fused extension, round 6).
//
// The decode layer's launch chain (llama_w4a8_unpad.py:406-438) has two 1 -> N edges: input_layernorm -> qkv_proj and
// post_attention_layernorm -> gate_up_proj.  As two launches the GEMV cannot ask for a single weight byte before the row
// kernel -- a pure latency chain on 16 CUs -- has ended, and the row kernel cannot hide behind anything.  Here:
//
//   workgroups [0, M)            the add + norm + quant ROWS (row_kernels.h: the very row body of the stand-alone kernels, 256
//                                threads per row), codes / scale / sum stored write-through (sc1), one arrival per row on `ready`;
//   workgroups [M, M + groups)   the GEMV's tiles: 4 waves = the four K parts of one 64-channel group (gate | up tile-row pair
//                                with the SiLU epilogue).  Every wave requests its WHOLE weight part at once -- a 16-step
//                                register ring = 32 KiB per wave, i.e. every byte of the projection is in flight a few hundred
//                                nanoseconds into the launch -- and only then waits for `ready` (one wave polls one word with
//                                agent-scope loads, the others park on s_barrier).  Behind the gate: activation codes with sc1
//                                loads -> LDS in the weights' k order -> 16 steps of unpack + MFMA on weights that are already
//                                in registers -> K-part exchange -> epilogue.
//
// The HBM stream of the projection (58.7 MB for Llama-3-8B's gate_up: 10.3 us at the chip's 5.7 TB/s) and the row chain (6-8 us)
// overlap completely; what is left behind the gate is the activation round trip and ~1 us of arithmetic.
//
// Deadlock freedom does not depend on dispatch order: the host only launches when EVERY workgroup of the grid can be resident at
// once (occupancy query x CUs >= grid), so the rows always get their CUs; the poll is bounded and a give-up poisons the outputs
// (NaN) and raises the caller's sticky error word.  `ready` is zeroed by the caller before the launch (the decode step's first
// kernel zeroes the step's words; not by this kernel: a workgroup that starts late must not see a half-reset counter).
// Arithmetic: the row body and the GEMV epilogue are the stand-alone kernels' code (same operand order, -ffp-contract=off):
// codes, scales, sums, residual and outputs are bit-identical to omni_splitk_add_rms_norm_general_fuse_sum followed by
// omni_w4a8_per_chn_gemm / omni_w4a8_per_chn_gemm_silu (tests/test_norm_gemv_fused_gpu.py).
#pragma once
#include "qgemm_kernel.h"
#include "row_kernels.h"

namespace omni {

constexpr int NGF_KW = 4;          // K parts (waves) per GEMV workgroup
constexpr int NGF_RING = 16;       // k-steps a wave holds in registers = its whole part (K <= 4 * 16 * 64)
constexpr int NGF_AR = 8;          // k-steps per activation round (one LDS buffer of 16 rows x 512 B)
constexpr int NGF_THREADS = 64 * NGF_KW;
constexpr int NGF_CLK_MARKS = 8;
constexpr int NGF_SYNC_WORDS = 32; // per call site: [0] arrivals, [16 .. 32) {scale, sum} pairs of the rows

struct NormGemvArgs {
  // ---- rows ----
  const half_t* gamma;      // [K]
  int8_t* codes;            // [M, K] out (and the hand-off buffer)
  half_t* sum_out;          // [M] (NULL with FUSE_SUM = false)
  half_t* scale_out;        // [M]
  float eps;
  int nv;                   // virtual threads of the reference's row reduction
  // ---- hand-off ----
  uint32_t* sync;           // [NGF_SYNC_WORDS], zeroed by the caller before the launch
  uint32_t* err;            // sticky error word (never cleared here)
  // ---- GEMV ----
  const uint8_t* W;         // packed [N, K/2]
  const uint8_t* s2s;       // per-group: [K/128, N] second-level scales / zeros
  const uint8_t* s2z;
  const half_t* wscales;    // [N]
  const half_t* wsz;        // [N] (per-channel)
  half_t* out;              // [M, out_stride]  (SiLU form: act [M, N/2])
  uint32_t* amax;           // SiLU form: row-maximum slots (common.h)
  long long out_stride;
  int M, N, K;
  unsigned long long* clk;  // optional [grid][NGF_CLK_MARKS] 100 MHz wall-clock marks (timeline probe)
};

#define NGF_CLK(k)                                                                                          \
  do {                                                                                                      \
    if (a.clk && threadIdx.x == 0) a.clk[(size_t)blockIdx.x * NGF_CLK_MARKS + (k)] = wall_clock64();        \
  } while (0)

__device__ __forceinline__ uint32_t ngf_ld_agent(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16 bytes with ONE agent-scope load (buffer_load_dwordx4 ... sc1: bypasses this CU's L1; the producer stored write-through).
// A raw buffer load, because hipcc has no 16-byte atomic load and counts the builtin like any other load (aux 16 = sc1).
__device__ __forceinline__ uint4 ngf_ld16_agent(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off) {
  const v4i v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 16);
  return make_uint4((uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]);
}

// sink of the row body: codes written through (8-byte agent-scope stores), scale / sum kept for the pair word
struct SinkHandoff {
  int8_t* orow;
  half_t* sum_out;
  half_t* scale_out;
  int row;
  uint32_t scale_bits, sum_bits;     // (thread 0)
  __device__ __forceinline__ void codes(int i, uint2 c) const {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(orow + i), (unsigned long long)c.x | ((unsigned long long)c.y << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ void scale(half_t s) { scale_out[row] = s; scale_bits = __builtin_bit_cast(uint16_t, s); }
  __device__ __forceinline__ void sum(half_t s) { sum_out[row] = s; sum_bits = __builtin_bit_cast(uint16_t, s); }
};

// MODE: MODE_CHN / MODE_GRP.  EPI = 1: gate_up with silu_and_mul in the epilogue (w4a8_gemv_kernel's EPI form).
// RV: 8-element vectors per thread of a row (K <= 256 * RV * 8).  Src: the row's source (row_kernels.h / elementwise.hip).
template <int MODE, int EPI, int RV, bool FUSE_SUM, typename Src>
__global__ __launch_bounds__(NGF_THREADS, 2) void norm_gemv_fused_kernel(NormGemvArgs a, Src src0) {
  constexpr int KW = NGF_KW, RING = NGF_RING, AR = NGF_AR, MT = 16;
  constexpr int RK = AR * KSTEP;                          // 512 k per activation round
  constexpr int APT = (MT * RK / 16) / 64;                 // 8 16-B activation pieces per lane per round
  constexpr int CPR = RK / 256;                            // 16-piece chunks per activation row of a round
  static_assert(MODE == MODE_CHN || MODE == MODE_GRP, "int4 modes");
  __shared__ __attribute__((aligned(16))) uint8_t lds_all[KW][2][MT * RK];      // 64 KiB; rows: the f32 copy of the row
  __shared__ float red[96];
  __shared__ uint32_t s_fail;
  src0.pin();
  asm volatile("" ::"s"(a.gamma), "s"(a.codes), "s"(a.sync), "s"(a.W), "s"(a.wscales), "s"(a.wsz), "s"(a.out), "s"(a.M), "s"(a.N),
               "s"(a.K), "s"(a.clk));
  NGF_CLK(0);

  // =============================== rows =====================================================================
  if ((int)blockIdx.x < a.M) {
    const int row = blockIdx.x;
    const Src src = src0.at_row(row);
    SinkHandoff sink{a.codes + (size_t)row * a.K, a.sum_out, a.scale_out, row, 0u, 0u};
    struct Hook {
      const NormGemvArgs& a;
      __device__ __forceinline__ void loaded() const { NGF_CLK(5); }
    };
    general_norm_v2_row<NGF_THREADS, RV, FUSE_SUM, Src, SinkHandoff, Hook>(src, a.gamma, sink, a.eps, a.K, a.nv,
                                                                           reinterpret_cast<float*>(&lds_all[0][0][0]), red, Hook{a});
    NGF_CLK(1);
    if (threadIdx.x == 0)
      __hip_atomic_store(a.sync + 16 + row, sink.scale_bits | (sink.sum_bits << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave: its write-through stores have landed
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    NGF_CLK(2);
    return;
  }

  // =============================== GEMV tiles ===================================================================
  const int lane = threadIdx.x & 63;
  const int kw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ng = (int)blockIdx.x - a.M;                   // 64-channel group (EPI: gate | up tile-row pair)
  const int kpart = a.K / KW;
  const int k_begin = kw * kpart;
  const int nsteps = kpart / KSTEP;                       // <= RING (host)
  uint8_t (*lds)[MT * RK] = lds_all[kw];

  const int lx = (lane >> 3) & 1, lc = lane & 7, le = lane >> 4;
  const int trow = EPI == 1 ? (lx ? a.N / 64 + ng : ng) : 2 * ng + lx;
  const uint8_t* wbase = a.W + ((size_t)trow * (a.K / 32)) * 512 + (lc * 4 + le) * 16;
  const size_t gcol = (size_t)trow * 32 + lc * 4;

  // ---- the whole weight part of this wave: 2 x 16-B non-temporal loads per k-step, all in flight -------------
  uint4 wq[RING][2];
  uint32_t gs[RING / 2], gz[RING / 2];                    // per-group: one parameter dword per 128 k
#pragma unroll
  for (int s = 0; s < RING; ++s) {
    const int ks = s < nsteps ? s : 0;                    // (short parts re-read step 0; never consumed)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const v4i v = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(wbase + (size_t)((k_begin + ks * KSTEP) / 32 + j) * 512));
      wq[s][j] = make_uint4((uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]);
    }
    if constexpr (MODE == MODE_GRP) {
      if ((s & 1) == 0) {
        const size_t off = (size_t)((k_begin + ks * KSTEP) / 128) * a.N + gcol;
        gs[s / 2] = *reinterpret_cast<const uint32_t*>(a.s2s + off);
        gz[s / 2] = *reinterpret_cast<const uint32_t*>(a.s2z + off);
      }
    }
  }
  // epilogue operands of the row block this wave finishes (ab = kw)
  const int mcol = lane & 15;
  const int i0 = (lane >> 4) * 4;
  auto chan = [&](int ab) -> int {
    if constexpr (EPI == 1) return (i0 >> 3) * (a.N / 2) + ng * 32 + ab * 8 + (i0 & 7);   // gate | up channel
    else return ng * 64 + (i0 >> 3) * 32 + ab * 8 + (i0 & 7);
  };
  uint2 swv = *reinterpret_cast<const uint2*>(a.wscales + chan(kw));
  uint2 szv = make_uint2(0u, 0u);
  if constexpr (MODE == MODE_CHN) szv = *reinterpret_cast<const uint2*>(a.wsz + chan(kw));
  NGF_CLK(1);

  // ---- the gate: all rows published ---------------------------------------------------------------------
  if (threadIdx.x == 0) s_fail = 0u;
  if (kw == 0) {
    bool ok = false;
    for (int spin = 0; spin < (1 << 21); ++spin) {
      if ((int32_t)(ngf_ld_agent(a.sync) - (uint32_t)a.M) >= 0) { ok = true; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    if (!ok && lane == 0) {      // a row never arrived: report and poison instead of hanging the queue
      __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_fail = 1u;
    }
  }
  __syncthreads();
  NGF_CLK(2);
  const uint32_t pairw = ngf_ld_agent(a.sync + 16 + (mcol < a.M ? mcol : a.M - 1));

  // ---- activation rounds: codes (sc1 loads) -> LDS in the weights' k order (w4a8_gemv_kernel's image).  BOTH rounds are
  //      requested at once (each one after the other cost a second exposed round trip behind the gate: timeline r06_b) ----
  const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(a.codes, 0, a.M * a.K, 0x00020000);
  uint4 areg[2][APT];
  auto piece = [&](int j, int& m, int& kk) {
    m = 4 * (j / CPR) + (lane >> 4);
    kk = (lane & 15) + 16 * (j % CPR);
  };
  auto load_a = [&](int r, int kr) {
#pragma unroll
    for (int j = 0; j < APT; ++j) {
      int m, kk;
      piece(j, m, kk);
      const int mc = m < a.M ? m : a.M - 1;               // rows >= M re-read the last row (never stored)
      const int k = kr + kk * 16;
      areg[r][j] = ngf_ld16_agent(crsrc, (uint32_t)(mc * a.K + (k < a.K ? k : 0)));
    }
  };
  auto store_a = [&](int r) {
#pragma unroll
    for (int j = 0; j < APT; ++j) {
      int m, kk;
      piece(j, m, kk);
      const int kp = kk >> 2, tp = (kk >> 1) & 1, d = kk & 1;
      uint8_t* dst = &lds[r][(kp * MT + m) * 64 + tp * 8 + d * 4];
      *reinterpret_cast<uint32_t*>(dst + ((0 + kp) & 3) * 16) = areg[r][j].x;
      *reinterpret_cast<uint32_t*>(dst + ((1 + kp) & 3) * 16) = areg[r][j].y;
      *reinterpret_cast<uint32_t*>(dst + ((2 + kp) & 3) * 16) = areg[r][j].z;
      *reinterpret_cast<uint32_t*>(dst + ((3 + kp) & 3) * 16) = areg[r][j].w;
    }
  };
  v4i acc[4];
#pragma unroll
  for (int ab = 0; ab < 4; ++ab) acc[ab] = (v4i){0, 0, 0, 0};
  auto step = [&](int s, const uint8_t* abuf) {           // s: step inside the ring (compile-time after unrolling)
    const uint4 t0 = wq[s][0], t1 = wq[s][1];
    const uint32_t dw[2][4] = {{t0.x, t0.z, t1.x, t1.z}, {t0.y, t0.w, t1.y, t1.w}};
    v4i wa[4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        uint32_t u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q] = (dw[y][q] >> (4 * x)) & 0x0F0F0F0Fu;
        if constexpr (MODE == MODE_GRP) {
          const uint32_t sc = (gs[s / 2] >> (8 * (x * 2 + y))) & 0xFFu;
#pragma unroll
          for (int q = 0; q < 4; ++q) u[q] = u[q] * sc;
          vadd4_zbyte_x4(u, gz[s / 2], x * 2 + y);
        }
        wa[x * 2 + y] = (v4i){(int)u[0], (int)u[1], (int)u[2], (int)u[3]};
      }
    const int sa = s % AR;
    const int pos = ((lane >> 4) + sa) & 3;
    const v4i bf = *reinterpret_cast<const v4i*>(abuf + ((sa * MT + (lane & 15)) * 4 + pos) * 16);
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) acc[ab] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[ab], bf, acc[ab], 0, 0, 0);
  };
  load_a(0, k_begin);
  if (nsteps > AR) load_a(1, k_begin + RK);
  store_a(0);
  if (nsteps > AR) store_a(1);
#pragma unroll
  for (int s = 0; s < AR; ++s)
    if (s < nsteps) step(s, lds[0]);
  if (nsteps > AR) {
#pragma unroll
    for (int s = AR; s < RING; ++s)
      if (s < nsteps) step(s, lds[1]);
  }
  NGF_CLK(3);

  // ---- combine the K parts, epilogue (w4a8_gemv_kernel's write-back, KW = 4: wave kw finishes row block ab = kw) -----
  static_assert(4 * 64 * 16 <= 2 * MT * RK, "partials fit the wave's staging buffers");
  v4i* mine = reinterpret_cast<v4i*>(&lds_all[kw][0][0]);
#pragma unroll
  for (int ab = 0; ab < 4; ++ab) mine[ab * 64 + lane] = acc[ab];
  __syncthreads();
  const bool failed = s_fail != 0u;
  float rowmax = 0.0f;
  const int m = mcol;
  {
    v4i a4 = (v4i){0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < KW; ++w) a4 += reinterpret_cast<const v4i*>(&lds_all[w][0][0])[kw * 64 + lane];
    typedef _Float16 v4h_t __attribute__((ext_vector_type(4)));
    const v4h_t sw4 = __builtin_bit_cast(v4h_t, swv);
    v4h_t sz4 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
    if constexpr (MODE == MODE_CHN) sz4 = __builtin_bit_cast(v4h_t, szv);
    const float sa = (float)__builtin_bit_cast(half_t, (uint16_t)(pairw & 0xFFFFu));
    float as = 0.f;
    if constexpr (MODE == MODE_CHN) as = (float)__builtin_bit_cast(half_t, (uint16_t)(pairw >> 16));
    half_t o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = epilogue<MODE>(a4[r], (float)sw4[r], sa, (float)sz4[r], as);
    if (failed) {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = __builtin_bit_cast(half_t, (uint16_t)0x7E00);
    }
    if constexpr (EPI == 1) {
      // lanes < 32 hold the fp16 gate outputs, lanes >= 32 the up outputs of the same (row, 4 channels)
      const uint2 me = *reinterpret_cast<const uint2*>(o);
      const uint2 other = make_uint2((uint32_t)__shfl_xor((int)me.x, 32, 64), (uint32_t)__shfl_xor((int)me.y, 32, 64));
      const uint2 g2 = lane < 32 ? me : other, u2 = lane < 32 ? other : me;
      const v4h_t g4 = __builtin_bit_cast(v4h_t, g2), u4 = __builtin_bit_cast(v4h_t, u2);
      half_t act[4];
      float mx = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        act[r] = silu_mul_h(g4[r], u4[r]);
        mx = __builtin_fmaxf(mx, __builtin_fabsf((float)act[r]));
      }
      rowmax = mx;
      if (lane < 32 && m < a.M)
        *reinterpret_cast<uint2*>(a.out + (size_t)m * a.out_stride + (ng * 32 + kw * 8 + (i0 & 7))) = *reinterpret_cast<const uint2*>(act);
    } else {
      if (m < a.M) *reinterpret_cast<uint2*>(a.out + (size_t)m * a.out_stride + chan(kw)) = *reinterpret_cast<const uint2*>(o);
    }
  }
  if constexpr (EPI == 1) {
    // row maxima: across the lanes of a row (lane & 15), then across the K-part waves, then one atomic per row
    __shared__ float smax[KW][MT];
    const float v = rows4_max(rowmax);
    if (lane < 16) smax[kw][lane] = lane < a.M ? v : 0.0f;
    __syncthreads();
    if (threadIdx.x < MT) {
      float r = smax[0][threadIdx.x];
#pragma unroll
      for (int w = 1; w < KW; ++w) r = __builtin_fmaxf(r, smax[w][threadIdx.x]);
      if ((int)threadIdx.x < a.M) amax_raise(a.amax, threadIdx.x, ng >> 3, r);
    }
  }
  NGF_CLK(4);
}

}  // namespace omni
