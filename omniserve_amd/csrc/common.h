// Shared device helpers for the OmniServe MI355X (gfx950) hot-path kernels.
// wave = 64 lanes everywhere; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace omni {

template <int V> struct IntTag { static constexpr int value = V; };      // static index handed to a generic lambda

typedef _Float16 half_t;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

#define OMNI_OK 0
#define OMNI_EINVAL (-22)
#define OMNI_ENOMEM (-12)
#define OMNI_ELAUNCH (-5)

// Environment-driven A/B knobs (planner thresholds, debug ablations) exist only in tuning builds
// (tools/build_variant.sh NAME "-DOMNI_TUNING ..."): the release library never calls getenv and has no "wrong results" switch.
#ifdef OMNI_TUNING
#include <cstdlib>
static inline int omni_knob(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
#else
#define omni_knob(name, dflt) (dflt)
#endif

static inline int omni_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? OMNI_OK : OMNI_ELAUNCH;
}

// Keeps an f32 intermediate as a materialised, once-rounded f32 value.  Without it the backend
// folds `(half)(a * (float)b_half)` into v_fma_mixlo_f16 (one rounding straight to fp16), whereas
// the reference (and oracle/) round to f32 first and to fp16 second; the two differ in ~2^-13 of
// the elements.
__device__ __forceinline__ float rounded_f32(float x) {
  asm volatile("" : "+v"(x));
  return x;
}

// cvt.rni.sat.s8.f32 (reference kernels/csrc/utils.cuh:79-84): round half to
// even, saturate to [-128,127], NaN -> 0.
// load through a pointer that came out of an int64 page table: tell the compiler it is global memory, otherwise
// it emits flat_load (slower path, and it also counts in lgkmcnt, tying LDS waits to outstanding memory loads)
template <class T>
__device__ __forceinline__ T gload(const void* p) {
  return *(const __attribute__((address_space(1))) T*)(p);
}
// 16 bytes per lane straight from global memory into LDS (gfx950 global_load_lds_dwordx4).  `lds_wave_base` must be
// wave-uniform: lane l lands at lds_wave_base + 16*l.  Completion is tracked by vmcnt.
__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The same DMA hidden from the compiler (inline asm): hipcc counts a pending global_load_lds as an LDS event of a
// second kind and then waits lgkmcnt(0) in front of EVERY consumer of a ds_read while one is in flight (read -> full
// wait -> MFMA, no LDS pipelining at all).  The caller owns the completion: lds_dma_wait_all() before the barrier that
// publishes the tile.  M0 is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void lds_dma16_untracked(const void* gsrc, void* lds_wave_base) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(
      (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)lds_wave_base);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}
__device__ __forceinline__ void lds_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

typedef uint32_t v4u_native __attribute__((ext_vector_type(4)));
template <>
__device__ __forceinline__ uint4 gload<uint4>(const void* p) {
  const v4u_native t = *(const __attribute__((address_space(1))) v4u_native*)(p);
  return make_uint4(t[0], t[1], t[2], t[3]);
}
template <class T>
__device__ __forceinline__ void gstore(void* p, T v) {
  *(__attribute__((address_space(1))) T*)(p) = v;
}

__device__ __forceinline__ int8_t rni_sat_s8(float x) {
  float r = __builtin_rintf(x);
  r = (r != r) ? 0.0f : r;
  r = __builtin_fminf(__builtin_fmaxf(r, -128.0f), 127.0f);
  return (int8_t)(int)r;
}

// silu(x) * y of silu_and_mul (activation_kernels.cu:10-13,84-97): silu in f32, rounded to fp16, times y in f32, rounded
// to fp16.  The reference is built with --use_fast_math (ex2.approx + approximate division): v_exp_f32 / v_rcp_f32 are
// the same class of approximation.  ONE definition for every kernel that produces this value (silu_and_mul, the fused
// SiLU quantiser, the gate_up GEMV's fused epilogue), so they agree bit for bit.
__device__ __forceinline__ half_t silu_mul_h(half_t a, half_t b) {
  const float xf = (float)a;
  const float e = __builtin_amdgcn_exp2f(xf * -1.4426950408889634f);
  const half_t s = (half_t)(xf * __builtin_amdgcn_rcpf(1.0f + e));
  return (half_t)((float)s * (float)b);
}

// four fp16 values (two packed dwords) -> the four int8 codes rni_sat_s8(f32(x) * q), packed little-endian, for a row
// whose multiplier is q = 127 / max|x| (quant_multiplier below).  2.25 VALU per element instead of 5.75 for the
// literal rint / NaN test / clamp / convert / pack sequence:
//   * f32(x) * q is one rounding to f32, as in rni_sat_s8(x * q);
//   * adding 1.5 * 2^23 rounds that product to an integer half-to-even (the f32 add IS the rounding) and leaves its
//     two's-complement byte in the low mantissa bits;
//   * |x| <= max|x| makes |x * q| <= 127 (1 + 2^-23): the saturation can never fire, no clamp;
//   * NaN products (NaN or inf inputs: inf * 0) keep a NaN through the add, and every NaN this path can produce --
//     the default NaN and widened fp16 payloads -- has a zero low byte: code 0, which is cvt.rni.sat's NaN -> 0.
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float quant_multiplier(float amax) {
  // 127 / amax; an all-zero row (amax = 0: the reference divides by zero, then 0 * inf = NaN -> code 0) gets 0 so that the
  // products are 0 -> code 0 without going through NaN
  return amax == 0.0f ? 0.0f : 127.0f / amax;
}
// f32(x) * q in ONE instruction (v_fma_mix_f32 widens the fp16 operand itself; fma(x, q, +0) is the once-rounded product,
// i.e. v_cvt_f32_f16 + v_mul_f32) -- hipcc emits shift + convert + multiply; and the packed f32 add as written (left to
// itself the backend splits most of the pairs again).
__device__ __forceinline__ float mul_f16lo_f32(uint32_t h2, float q) {
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(q));
  return r;
}
__device__ __forceinline__ float mul_f16hi_f32(uint32_t h2, float q) {
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(q));
  return r;
}
// (64-bit integer operands: with <2 x float> asm outputs hipcc 7.2 read element 0 for both elements of the result)
__device__ __forceinline__ uint64_t pk_add_f32_bits(uint64_t a, uint64_t b) {
  uint64_t r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t quant4_f16(uint32_t h01, uint32_t h23, float q) {
  constexpr uint64_t magic = 0x4B4000004B400000ull;       // {1.5 * 2^23, 1.5 * 2^23}
  auto pair = [](float lo, float hi) -> uint64_t {
    return ((uint64_t)__builtin_bit_cast(uint32_t, hi) << 32) | __builtin_bit_cast(uint32_t, lo);
  };
  const uint64_t pa = pk_add_f32_bits(pair(mul_f16lo_f32(h01, q), mul_f16hi_f32(h01, q)), magic);
  const uint64_t pb = pk_add_f32_bits(pair(mul_f16lo_f32(h23, q), mul_f16hi_f32(h23, q)), magic);
  // v_perm_b32(S0, S1, sel): selector byte 0-3 -> byte of S1, 4-7 -> byte of S0, 0x0c -> 0x00
  const uint32_t lo = __builtin_amdgcn_perm((uint32_t)(pa >> 32), (uint32_t)pa, 0x0c0c0400u);
  const uint32_t hi = __builtin_amdgcn_perm((uint32_t)(pb >> 32), (uint32_t)pb, 0x04000c0cu);
  return lo | hi;
}

// cvt.rni.sat.u8.f32
__device__ __forceinline__ uint32_t rni_sat_u8(float x) {
  float r = __builtin_rintf(x);
  r = (r != r) ? 0.0f : r;
  r = __builtin_fminf(__builtin_fmaxf(r, 0.0f), 255.0f);
  return (uint32_t)(int)r;
}

// ---- reference block-reduction trees (kernels/csrc/reduction_utils.cuh:25-164)
// The reference reduces with a 32-lane xor butterfly (masks 16,8,4,2,1) and then
// a butterfly over the zero-padded 32 warp partials.  On wave64 the same masks
// stay inside each 32-lane half, so the float summation tree is reproduced
// exactly: `virtual warp` = 32-lane half of a wave.
__device__ __forceinline__ float half_wave_butterfly_sum(float v) {
#pragma unroll
  for (int mask = 16; mask > 0; mask >>= 1) v = v + __shfl_xor(v, mask, 64);
  return v;
}
__device__ __forceinline__ float half_wave_butterfly_max(float v) {
#pragma unroll
  for (int mask = 16; mask > 0; mask >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, mask, 64));
  return v;
}

// Block-wide sum with the reference tree.  `red` is LDS scratch of >= 32 floats.
// blockDim.x must be a multiple of 32 and <= 1024.  All threads get the result.
__device__ __forceinline__ float ref_block_sum(float v, float* red) {
  const int lane32 = threadIdx.x & 31;
  const int vwarp = threadIdx.x >> 5;
  const int nwarps = blockDim.x >> 5;
  v = half_wave_butterfly_sum(v);
  __syncthreads();  // protect `red` against a previous use
  if (lane32 == 0) red[vwarp] = v;
  __syncthreads();
  float w = (lane32 < nwarps) ? red[lane32] : 0.0f;
  w = half_wave_butterfly_sum(w);
  return w;
}

__device__ __forceinline__ float ref_block_max(float v, float* red, float pad) {
  const int lane32 = threadIdx.x & 31;
  const int vwarp = threadIdx.x >> 5;
  const int nwarps = blockDim.x >> 5;
  v = half_wave_butterfly_max(v);
  __syncthreads();
  if (lane32 == 0) red[vwarp] = v;
  __syncthreads();
  float w = (lane32 < nwarps) ? red[lane32] : pad;
  w = half_wave_butterfly_max(w);
  return w;
}

// ---- DPP lane exchange (one VALU op, no LDS crossbar round trip like ds_bpermute / __shfl_xor) ----------
// CTRL: quad_perm = p0 | p1<<2 | p2<<4 | p3<<6; row_shl:n = 0x100+n; row_shr:n = 0x110+n; row_ror:n = 0x120+n;
// row_mirror = 0x140; row_half_mirror = 0x141.  A row = 16 lanes.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_xor1(float v) { return dpp_mov<0xB1>(v); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float lane_xor2(float v) { return dpp_mov<0x4E>(v); }   // quad_perm [2,3,0,1]

// 64-lane max / sum, result in every lane.  Pairing order: xor 1, 2 inside quads, mirrored halves and rows,
// then the four 16-lane rows through v_readlane.  max is exact in any order; wave_sum64 is only used where
// the summation order is not part of a parity contract (softmax denominators, q.k of the current token).
// Reductions over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) without LDS traffic: gfx950's
// v_permlane32_swap (upper half of a <-> lower half of b) and v_permlane16_swap (odd rows of a <-> even rows of b)
// on two copies of the value leave every lane holding both partners; a ds_bpermute costs an LDS round trip each.
__device__ __forceinline__ float rows4_max(float v) {
  unsigned x = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  v = __builtin_fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
  x = __builtin_bit_cast(unsigned, v);
  r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  return __builtin_fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float rows4_sum(float v) {   // (l + l^32) + (l^16 + l^48), the order of the shuffle tree
  unsigned x = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
  x = __builtin_bit_cast(unsigned, v);
  r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

__device__ __forceinline__ float wave_max64(float v) {
  v = __builtin_fmaxf(v, lane_xor1(v));
  v = __builtin_fmaxf(v, lane_xor2(v));
  v = __builtin_fmaxf(v, dpp_mov<0x141>(v));
  v = __builtin_fmaxf(v, dpp_mov<0x140>(v));
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return __builtin_fmaxf(__builtin_fmaxf(r0, r1), __builtin_fmaxf(r2, r3));
}
__device__ __forceinline__ float wave_sum64(float v) {
  v = v + lane_xor1(v);
  v = v + lane_xor2(v);
  v = v + dpp_mov<0x141>(v);
  v = v + dpp_mov<0x140>(v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}

// ---- row maxima handed from the producer of an fp16 activation to the projection that quantises it (fused extension) ----
// u32 [AMAX_XCC][AMAX_ROWS][AMAX_SUB] holding f32 bit patterns of candidates for max |x| of each activation row
// (<= 16 rows), zeroed by the caller before the producer runs.  A producer workgroup raises [its XCD][row][sub] with an
// integer max (non-negative f32 bit patterns order like unsigned integers: exact and order independent).  Every word is
// only ever touched from ONE XCD, so the read-modify-write can stay in that XCD's L2 (workgroup-scope atomic: no sc1, the
// L2 is the point of coherence for all CUs of an XCD) instead of going to the memory side, and 448 workgroups x 16 rows
// finishing together spread over 8 L2s x 4 cache lines each (the first version -- device-scope atomics on 128 words in 4
// cache lines -- cost the gate_up GEMV 5.5 us of its 15).  The kernel boundary publishes the words to the consumer.
constexpr int AMAX_XCC = 8, AMAX_ROWS = 16, AMAX_SUB = 8;
constexpr int AMAX_WORDS = AMAX_XCC * AMAX_ROWS * AMAX_SUB;      // 4 KiB per activation tensor

__device__ __forceinline__ int xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return (int)(v & (AMAX_XCC - 1));
}
__device__ __forceinline__ void amax_raise(uint32_t* amax, int row, int sub, float v) {
  uint32_t* w = amax + ((size_t)xcc_id() * AMAX_ROWS + row) * AMAX_SUB + (sub & (AMAX_SUB - 1));
  __hip_atomic_fetch_max(w, __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// wave-cooperative read: every lane gets max |x| of row (lane & 15).  Lane l reads the 8 candidates of its row from XCDs
// 2 * (l >> 4) and 2 * (l >> 4) + 1 (four 16-B loads), the four lanes of a row meet through the permlane swaps.
// Split into issue (loads only) and finish so that a caller can put other loads in flight before it waits.
struct AmaxRaw { uint4 a0, a1, b0, b1; };
__device__ __forceinline__ AmaxRaw amax_rows_issue(const uint32_t* amax) {
  const int lane = threadIdx.x & 63;
  const uint32_t* base = amax + ((size_t)(2 * (lane >> 4)) * AMAX_ROWS + (lane & 15)) * AMAX_SUB;
  AmaxRaw r;
  r.a0 = *reinterpret_cast<const uint4*>(base); r.a1 = *reinterpret_cast<const uint4*>(base + 4);
  r.b0 = *reinterpret_cast<const uint4*>(base + AMAX_ROWS * AMAX_SUB);
  r.b1 = *reinterpret_cast<const uint4*>(base + AMAX_ROWS * AMAX_SUB + 4);
  return r;
}
__device__ __forceinline__ float amax_rows_finish(const AmaxRaw& r) {
  const uint32_t ma = max(max(max(r.a0.x, r.a0.y), max(r.a0.z, r.a0.w)), max(max(r.a1.x, r.a1.y), max(r.a1.z, r.a1.w)));
  const uint32_t mb = max(max(max(r.b0.x, r.b0.y), max(r.b0.z, r.b0.w)), max(max(r.b1.x, r.b1.y), max(r.b1.z, r.b1.w)));
  return rows4_max(__builtin_bit_cast(float, max(ma, mb)));
}
__device__ __forceinline__ float amax_rows_wave(const uint32_t* amax) { return amax_rows_finish(amax_rows_issue(amax)); }

// ---- L2 weight prefetch riding on a latency-bound row kernel (fused extension) -------------------
// The decode step alternates bandwidth-bound GEMVs with row kernels (norm / quant: 16 workgroups, pure latency
// chains) during which HBM idles.  A row kernel launched with `blocks` extra workgroups uses them to pull the
// packed weights of the NEXT GEMV into the XCD-private L2s: the GEMV then finds the head of every wave's weight
// stream on chip (L2 latency instead of HBM latency) and only streams the rest.  Purely a hint: results never
// depend on it.  Geometry = the GEMV's own grid: workgroup (x, y) of a gx x gy grid streams `rows_per_wg`
// consecutive rows (of row_bytes each) starting at row x*rows_per_wg, K-slice y of gy, split over kw waves, i.e.
// parts [y*kw, (y+1)*kw) of each row; the leading pf_bytes of every part are fetched.  Workgroup id -> XCD is
// the dispatcher's observed round-robin (id % 8), both for the GEMV's workgroups and for the fetching ones.
struct PrefetchArgs {
  const uint8_t* base;
  long long row_bytes;
  int rows_per_wg, gx, gy, kw;
  int rows_pf;        // rows of a workgroup that are fetched (the first ones; <= rows_per_wg)
  int pf_bytes;       // multiple of 1024, <= row_bytes / (gy * kw)
  int blocks;         // extra workgroups (0 = off)
  int first_block;    // their first blockIdx.x (= number of row workgroups)
  int delay;          // s_sleep(16) iterations before the first fetch (tuning knob, normally 0)
};

// body of a fetching workgroup; `lds_scratch` >= 1 KiB per wave (LDS-DMA target, never read).
// No integer division inside the loops (gfx950 has none: each one is ~40 scalar instructions): a wave owns one
// (row, part) segment of a GEMV workgroup's bytes and walks its KiB pieces with a running pointer; with fewer
// segments than waves the waves of a segment interleave its pieces.  All waves advance together, so the first KiB of
// every part (what the GEMV's waves ask for first) still arrives first.
// `b`: the fetching workgroup's position in dispatch order (blockIdx.x of a 1-D grid; the linear id of a 3-D one)
__device__ __forceinline__ void prefetch_weights_to_l2(const PrefetchArgs& pf, void* lds_scratch, int b) {
  const int xcd = b & 7;
  const int b0 = pf.first_block + ((xcd - pf.first_block) & 7);        // first fetching workgroup on this XCD
  const int slot = (b - b0) >> 3;
  const int nslots = (pf.first_block + pf.blocks - b0 + 7) >> 3;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6, lane = threadIdx.x & 63;
  const int kib = pf.pf_bytes >> 10;                                   // 1-KiB pieces per part
  const int segs = pf.rows_pf * pf.kw;                                 // (row, part) pairs of one GEMV workgroup
  const long long part_bytes = pf.row_bytes / ((long long)pf.gy * pf.kw);
  // wave -> (first segment, segment stride, first piece, piece stride)
  int seg0, seg_step, piece0, piece_step;
  if (segs >= nwaves) { seg0 = wave; seg_step = nwaves; piece0 = 0; piece_step = 1; }
  else { const int per = nwaves / segs; seg0 = wave % segs; seg_step = segs; piece0 = wave / segs; piece_step = per;
         if (piece0 >= per) piece0 = kib; }                             // left-over waves (nwaves % segs) idle
  for (int d = 0; d < pf.delay; ++d) __builtin_amdgcn_s_sleep(16);     // let the row workgroups' own loads go first
  uint8_t* dst = reinterpret_cast<uint8_t*>(lds_scratch) + wave * 1024;
  const int items = pf.gx * pf.gy;
  for (int it = xcd + 8 * slot; it < items; it += 8 * nslots) {
    const int y = it / pf.gx, x = it - y * pf.gx;                      // one division per GEMV workgroup
    const uint8_t* wg = pf.base + (long long)x * pf.rows_per_wg * pf.row_bytes + (long long)y * pf.kw * part_bytes;
    for (int seg = seg0; seg < segs; seg += seg_step) {
      const int row = seg / pf.kw, part = seg - row * pf.kw;
      const uint8_t* src = wg + (long long)row * pf.row_bytes + (long long)part * part_bytes + lane * 16 +
                           (long long)piece0 * 1024;
      for (int piece = piece0; piece < kib; piece += piece_step, src += (long long)piece_step * 1024)
        lds_dma16(src, dst);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void prefetch_weights_to_l2(const PrefetchArgs& pf, void* lds_scratch) {
  prefetch_weights_to_l2(pf, lds_scratch, (int)blockIdx.x);
}

// Phase clocks for latency debugging (tools/phase_clocks.py; build with OMNI_HIPCC_EXTRA=-DOMNI_DEBUG_CLOCKS).
// Workgroup (0,0,0) thread 0 records the 100 MHz wall clock after draining its outstanding memory operations.
#ifdef OMNI_DEBUG_CLOCKS
static __device__ unsigned long long omni_dbg_clk[32];
#define OMNI_CLK(k)                                                                          \
  do {                                                                                       \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                              \
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)           \
      omni_dbg_clk[k] = wall_clock64();                                                      \
  } while (0)
#define OMNI_CLK_READER(name)                                                                \
  extern "C" int name(unsigned long long* out32) {                                           \
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(omni::omni_dbg_clk), 32 * sizeof(unsigned long long)) == hipSuccess ? 0 : -5; \
  }
#else
#define OMNI_CLK(k) do {} while (0)
#define OMNI_CLK_READER(name)
#endif

}  // namespace omni
