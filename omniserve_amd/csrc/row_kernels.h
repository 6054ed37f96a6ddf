// Row-kernel pieces shared between elementwise.hip (the stand-alone row kernels) and tools/experiments/mlp_fused.hip (the persistent MLP
// launch, whose norm "service" workgroups run the same row code): the split-K slab source and the general-norm row body.
// Arithmetic, reduction geometry and rounding points are documented in elementwise.hip / row_reduce.h; nothing here changes them.
#pragma once
#include "common.h"
#include "row_reduce.h"

namespace omni {

// ZP = true: the per-channel W4A8 epilogue ((acc*sw)*sa) - (sz*asum); ZP = false: the W8A8 / per-group one acc*(sw*sa)
// BATCH_: all of a thread's vectors requested back to back before any arithmetic (76 registers per vector: the stand-alone
// 4096-column row kernels run one vector per thread and leave it off; the 8192-column ones and the fused MLP launch's
// 256-thread service rows hold two)
template <bool ZP, bool BATCH_ = false>
struct SrcSlabAddT {  // residual += h(GEMM epilogue(sum of split-K slabs)), in place
  static constexpr bool BATCH = BATCH_;
  struct Raw { v4i t0[8], t1[8]; v8h a, sw, sz; };   // raw slab vectors: fetch() only requests, finish() sums (exact in any order)
  half_t* res;
  const int32_t* slab;    // [sk][M][N]
  size_t sstride;         // M*N
  int sk, stride;         // stride = N = hidden
  const half_t* wscales;  // [N]
  const half_t* wsz;      // [N]
  const half_t* ascales;  // [M] scales / sums of the GEMM's int8 input
  const half_t* asum;
  half_t sa_h, as_h;      // the row's activation scale / sum: REQUESTED by at_row, converted where finish() uses them (a
                          // conversion in at_row made hipcc wait for these two loads -- a full memory round trip with
                          // vmcnt(0) -- before it issued the row's slab / residual loads: ISA of round 4, profiles/r04_e)
  __device__ __forceinline__ void pin() const {   // kernel arguments of the row path, requested with a rider test's (elementwise.hip)
    asm volatile("" ::"s"(res), "s"(slab), "s"(sstride), "s"(sk), "s"(stride), "s"(wscales), "s"(wsz), "s"(ascales), "s"(asum));
  }
  __device__ __forceinline__ SrcSlabAddT at_row(int m) const {
    SrcSlabAddT r = *this;
    r.res = res + (size_t)m * stride;
    r.slab = slab + (size_t)m * stride;
    r.sa_h = ascales[m];
    if constexpr (ZP) r.as_h = asum[m];
    return r;
  }
  __device__ __forceinline__ void fetch(int i, Raw& r) const {   // requests only: the sums wait in finish()
    r.a = *reinterpret_cast<const v8h*>(res + i);
    r.sw = *reinterpret_cast<const v8h*>(wscales + i);
    if constexpr (ZP) r.sz = *reinterpret_cast<const v8h*>(wsz + i);
    // up to 8 slabs: all loads in flight at once (slabs beyond sk re-read slab 0 and are dropped)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t off = (size_t)(k < sk ? k : 0) * sstride + i;
      r.t0[k] = *reinterpret_cast<const v4i*>(slab + off);
      r.t1[k] = *reinterpret_cast<const v4i*>(slab + off + 4);
    }
  }
  __device__ __forceinline__ void finish(int i, const Raw& r, float (&x)[VT]) const {
    v8h o;
    const float sa = (float)sa_h;
    float as = 0.0f;
    if constexpr (ZP) as = (float)as_h;
    v4i s0 = (v4i){0, 0, 0, 0}, s1 = s0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s0 += k < sk ? r.t0[k] : (v4i){0, 0, 0, 0};
      s1 += k < sk ? r.t1[k] : (v4i){0, 0, 0, 0};
    }
    // slabs beyond 8 (the W8A8 down projection of the LServe driver leaves 14): further batches of 8 -- a plain loop made
    // every slab a dependent round trip for the one workgroup of a batch-1 row.  Integer sums: any order is exact.
    for (int k0 = 8; k0 < sk; k0 += 8) {
      v4i t0[8], t1[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const size_t off = (size_t)(k0 + k < sk ? k0 + k : 0) * sstride + i;
        t0[k] = *reinterpret_cast<const v4i*>(slab + off);
        t1[k] = *reinterpret_cast<const v4i*>(slab + off + 4);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s0 += k0 + k < sk ? t0[k] : (v4i){0, 0, 0, 0};
        s1 += k0 + k < sk ? t1[k] : (v4i){0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int e = 0; e < VT; ++e) {
      const int acc = e < 4 ? s0[e] : s1[e - 4];
      half_t ep;                                               // = the GEMM's fp16 output (qgemm_kernel.h: epilogue<>)
      if constexpr (ZP) {
        float t = (float)acc * (float)r.sw[e];
        t = t * sa;
        const float c = (float)r.sz[e] * as;
        ep = (half_t)(t - c);
      } else {
        const float sc = (float)r.sw[e] * sa;
        ep = (half_t)((float)acc * sc);
      }
      o[e] = (half_t)((float)r.a[e] + (float)ep);
      x[e] = (float)o[e];
    }
    *reinterpret_cast<v8h*>(res + i) = o;
  }
};
typedef SrcSlabAddT<true> SrcSlabAddChn;
typedef SrcSlabAddT<false> SrcSlabAddW8;

// rms_norm_general[_fuse_sum] (+ fused residual sources): NV = roundup32(min(hidden,1024))
// COH = true (the in-launch merge of kv4_decode_flash_kernel<..., LASTM>): the partials were written by OTHER workgroups of
// the SAME launch with agent-scope (sc1, write-through) stores; they are read with agent-scope (sc1) loads -- 8-byte relaxed
// atomics, which the compiler tracks like any load -- that bypass this CU's L1 (MI355X_MICROARCH.md, "Correctness boundaries":
// sc1 stores AND sc1 loads; plain loads could hit a line an earlier merger of this CU pulled in before the neighbouring
// ticket group's splits had written their part of it: part_ml holds 8 B per (head, split), so groups share 128-B lines
// whenever nsplit % 4 != 0).  Same values, same arithmetic: bit-identical to the two-launch form.
template <bool COH>
struct SrcAttnMergeT {  // merged decode-attention output (flash-decoding partials) of one token: [Hq*128] fp16
  static constexpr bool BATCH = false;
  static constexpr int NS = 8;   // splits whose loads are issued together (further ones: plain loop)
  struct Raw { float m[NS], l[NS]; v4f a[NS], b[NS]; };
  const float* part_ml;   // [B,Hq,S,2]
  const float* part_o;    // [B,Hq,S,128]
  int nsplit, num_heads, token;
  __device__ __forceinline__ void pin() const { asm volatile("" ::"s"(part_ml), "s"(part_o), "s"(nsplit), "s"(num_heads)); }
  __device__ __forceinline__ SrcAttnMergeT at_row(int m) const { SrcAttnMergeT r = *this; r.token = m; return r; }
  static __device__ __forceinline__ float2 ld2(const float* q) {
    if constexpr (COH) {
      const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return make_float2(__builtin_bit_cast(float, (uint32_t)v), __builtin_bit_cast(float, (uint32_t)(v >> 32)));
    } else {
      return *reinterpret_cast<const float2*>(q);
    }
  }
  static __device__ __forceinline__ v4f ld4(const float* q) {
    if constexpr (COH) {
      const float2 lo = ld2(q), hi = ld2(q + 2);
      return (v4f){lo.x, lo.y, hi.x, hi.y};
    } else {
      return *reinterpret_cast<const v4f*>(q);
    }
  }
  __device__ __forceinline__ void fetch(int i, Raw& r) const {
    const size_t bh = (size_t)token * num_heads + (i >> 7);
    const int d = i & 127;
#pragma unroll
    for (int s = 0; s < NS; ++s) {   // branch-free: splits >= nsplit re-read split 0 and get weight 0
      const size_t pi = bh * nsplit + (s < nsplit ? s : 0);
      const float2 ml = ld2(part_ml + pi * 2);
      r.m[s] = s < nsplit ? ml.x : -1e30f;
      r.l[s] = ml.y;
      r.a[s] = ld4(part_o + pi * 128 + d);
      r.b[s] = ld4(part_o + pi * 128 + d + 4);
    }
  }
  __device__ __forceinline__ void finish(int i, const Raw& r, float (&x)[VT]) const {
    const size_t bh = (size_t)token * num_heads + (i >> 7);
    const int d = i & 127;
    // same operation order as kv4_decode_merge_kernel: max over s, then s ascending accumulation
    float M = -1e30f;
#pragma unroll
    for (int s = 0; s < NS; ++s) M = __builtin_fmaxf(M, r.m[s]);
    // (splits beyond the first NS: batches of NS whose loads are issued together -- a plain loop made every split a
    //  dependent memory round trip: 16 splits of the LServe sparse attention cost the merge 7.1 us instead of 4.7)
    for (int s0 = NS; s0 < nsplit; s0 += NS) {
      float mm[NS];
#pragma unroll
      for (int u = 0; u < NS; ++u) mm[u] = ld2(part_ml + (bh * nsplit + (s0 + u < nsplit ? s0 + u : 0)) * 2).x;
#pragma unroll
      for (int u = 0; u < NS; ++u) M = __builtin_fmaxf(M, s0 + u < nsplit ? mm[u] : -1e30f);
    }
    float l = 0.0f;
    float o[VT];
#pragma unroll
    for (int e = 0; e < VT; ++e) o[e] = 0.0f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s < nsplit) {
        const float w = __expf(r.m[s] - M);
        l += w * r.l[s];
#pragma unroll
        for (int e = 0; e < VT; ++e) o[e] += w * (e < 4 ? r.a[s][e] : r.b[s][e - 4]);
      }
    }
    for (int s0 = NS; s0 < nsplit; s0 += NS) {
      float2 ml[NS];
      v4f a[NS], b[NS];
#pragma unroll
      for (int u = 0; u < NS; ++u) {   // branch-free: splits >= nsplit re-read split 0 and are skipped below
        const size_t pi = bh * nsplit + (s0 + u < nsplit ? s0 + u : 0);
        ml[u] = ld2(part_ml + pi * 2);
        a[u] = ld4(part_o + pi * 128 + d);
        b[u] = ld4(part_o + pi * 128 + d + 4);
      }
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        if (s0 + u < nsplit) {
          const float w = __expf(ml[u].x - M);
          l += w * ml[u].y;
#pragma unroll
          for (int e = 0; e < VT; ++e) o[e] += w * (e < 4 ? a[u][e] : b[u][e - 4]);
        }
      }
    }
    const float inv = 1.0f / (l + 1e-6f);
#pragma unroll
    for (int e = 0; e < VT; ++e) x[e] = (float)(half_t)rounded_f32(o[e] * inv);   // = kv4_decode_merge_kernel's fp16 output
  }
};
typedef SrcAttnMergeT<false> SrcAttnMerge;

// `src`: the row's source (already at_row()); `sink`: where the row's int8 codes / fp16 scale / fp16 sum go (SinkGlobal:
// the row kernels' own outputs; the fused MLP launch parks the codes in LDS and publishes them in MFMA operand order).
struct NoRowHook { __device__ __forceinline__ void loaded() const {} };   // Hook::loaded(): the row's inputs are in (LDS copy written)
template <int RT, int RV, bool FUSE_SUM, typename Src, typename Sink, typename Hook = NoRowHook>
__device__ __forceinline__ void general_norm_v2_row(const Src& src, const half_t* __restrict__ gamma, Sink& sink,
                                                    float eps, int hidden, int nv, float* xs, float* red,
                                                    const Hook& hook = Hook()) {
  const int p = threadIdx.x;
  OMNI_CLK(0);
  float x[RV][VT];
  typename Src::Raw raw[Src::BATCH ? RV : 1];
  v8h g8[RV];   // gamma is requested with the inputs, not after the statistics
#pragma unroll
  for (int it = 0; it < RV; ++it) {
    const int i = (p + it * RT) * VT;
    const int ic = i < hidden ? i : 0;
    if constexpr (Src::BATCH) src.fetch(ic, raw[it]);
    g8[it] = *reinterpret_cast<const v8h*>(gamma + ic);
  }
#pragma unroll
  for (int it = 0; it < RV; ++it) {
    const int i = (p + it * RT) * VT;
    const bool ok = i < hidden;
    if (ok) {
      if constexpr (Src::BATCH) {
        src.finish(i, raw[it], x[it]);
      } else {
        src.fetch(i, raw[0]);
        src.finish(i, raw[0], x[it]);
      }
      *reinterpret_cast<v4f*>(xs + i) = (v4f){x[it][0], x[it][1], x[it][2], x[it][3]};
      *reinterpret_cast<v4f*>(xs + i + 4) = (v4f){x[it][4], x[it][5], x[it][6], x[it][7]};
    }
  }
  OMNI_CLK(1);
  hook.loaded();
  __syncthreads();
  float st[2][VT], tv[2];
  ordered_partials<2>(xs, p, nv, hidden, st, [](float (&v)[2][VT], int e, float val) {
    v[0][e] = v[0][e] + val;
    v[1][e] = v[1][e] + val * val;
  });
  OMNI_CLK(2);
  tree_sum8<2>(st, red, p, nv >> 5, tv);   // first barrier inside: everyone is done reading xs
  OMNI_CLK(3);
  const float mean = tv[0] / (float)hidden;
  const float rstd = 1.0f / __builtin_sqrtf(tv[1] / (float)hidden + eps);
  float amax_h = (float)(half_t)1e-6f;
#pragma unroll
  for (int it = 0; it < RV; ++it) {
    const int i = (p + it * RT) * VT;
    const bool ok = i < hidden;
    if (ok) {
      float yh[VT];
#pragma unroll
      for (int e = 0; e < VT; ++e) {
        float y = (x[it][e] - mean) * rstd;
        y = rounded_f32(y * (float)g8[it][e]);
        x[it][e] = y;
        yh[e] = (float)(half_t)y;
        amax_h = __builtin_fmaxf(amax_h, __builtin_fabsf(yh[e]));
      }
      if constexpr (FUSE_SUM) {
        *reinterpret_cast<v4f*>(xs + i) = (v4f){yh[0], yh[1], yh[2], yh[3]};
        *reinterpret_cast<v4f*>(xs + i + 4) = (v4f){yh[4], yh[5], yh[6], yh[7]};
      }
    }
  }
  OMNI_CLK(4);
  const float amax = block_max_rt<RT>(amax_h, red);   // barriers publish the fp16-rounded y in xs
  OMNI_CLK(5);
  // the codes only need the maximum: their stores go out first and drain while the fp16 sum is replayed
  if (p == 0) sink.scale((half_t)(amax / 127.0f));
  const float q = 127.0f / amax;
#pragma unroll
  for (int it = 0; it < RV; ++it) {
    const int i = (p + it * RT) * VT;
    if (i < hidden) sink.codes(i, pack8_i8(x[it], q));
  }
  OMNI_CLK(6);
  if constexpr (FUSE_SUM) {
    float hs[1][VT], tot[1];
    ordered_partials<1>(xs, p, nv, hidden, hs, [](float (&v)[1][VT], int e, float val) {
      v[0][e] = (float)(half_t)(v[0][e] + val);   // the reference accumulates this sum in fp16
    });
    tree_sum8<1>(hs, red, p, nv >> 5, tot);
    if (p == 0) sink.sum((half_t)tot[0]);
  }
  OMNI_CLK(7);
}

struct SinkGlobal {   // int8 codes of the row at `orow`, scale / sum of row `row` in [tokens] arrays
  int8_t* orow;
  half_t* sum_out;
  half_t* scale_out;
  int row;
  __device__ __forceinline__ void codes(int i, uint2 c) const { *reinterpret_cast<uint2*>(orow + i) = c; }
  __device__ __forceinline__ void scale(half_t s) const { scale_out[row] = s; }
  __device__ __forceinline__ void sum(half_t s) const { sum_out[row] = s; }
};

template <int RT, int RV, bool FUSE_SUM, typename Src>
__device__ __forceinline__ void general_norm_v2_body(int8_t* __restrict__ out, const Src& src0, const half_t* __restrict__ gamma,
                                                     half_t* __restrict__ sum_out, half_t* __restrict__ scale_out,
                                                     float eps, int hidden, int nv, float* xs, float* red) {
  const Src src = src0.at_row(blockIdx.x);
  SinkGlobal sink{out + (size_t)blockIdx.x * hidden, sum_out, scale_out, (int)blockIdx.x};
  general_norm_v2_row<RT, RV, FUSE_SUM, Src, SinkGlobal>(src, gamma, sink, eps, hidden, nv, xs, red);
}


}  // namespace omni
