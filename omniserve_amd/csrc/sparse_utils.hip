// LServe dynamic-sparsity helpers for MI355X (gfx950): paged min/max pooling of K (prefill) and the
// Quest-style page selector (decode).
//
// Replaces omniserve_backend.fused_attention_ctx_pool.paged_min_max_pool
//   (kernels/csrc/fused_attention/sparse_utils/ContextPool/context_pool_kernel.cu:17-95,145-213) and
// omniserve_backend.fused_attention_selector.single_query_page_selector
//   (sparse_utils/KVPageSelector/KVPageSelectorTemplate.hpp:482-493,1130-1252, fused_kv_page_selector.cpp:262-334).
//
// K page of a retrieval pool (H_r retrieval kv heads):
//   int4 data [H_r][tpb][64 B] | fp16 scale [H_r][tpb] | fp16 zero [H_r][tpb]
//   | fp16 kmax [tpb/sub][H_r][128] | fp16 kmin [tpb/sub][H_r][128]
#include "common.h"

namespace omni {

constexpr int SDH = 128;

struct PoolArgs {
  const half_t* k;            // [L, Hin, 128] contiguous (post-RoPE keys)
  const int64_t* kv_pointers; // retrieval table [B,2,max_blocks]
  const int* cu_seqlens;      // [B+1]
  const int* pooling_heads_idx;  // [pool_h] -> input head
  int max_blocks, num_input_heads, pool_h, pooling_size, page_size;
  int row_bytes;              // bytes of one token row of one head in the page: Dh/2 (KV4) or Dh (KV8)
};

// one workgroup per (page, sequence, pooled head); thread = one head dim, two halves of the page's sub-chunks
__global__ __launch_bounds__(256) void kv_min_max_pool_kernel(PoolArgs p) {
  const int page = blockIdx.x, b = blockIdx.y, r = blockIdx.z;
  const int begin = p.cu_seqlens[b], len = p.cu_seqlens[b + 1] - begin;
  if (page * p.page_size >= len) return;
  const int d = threadIdx.x & 127, half_id = threadIdx.x >> 7;
  const int hin = p.pooling_heads_idx[r];
  const int subs = p.page_size / p.pooling_size;
  uint8_t* pg = reinterpret_cast<uint8_t*>(p.kv_pointers[(size_t)b * 2 * p.max_blocks + page]);
  const size_t bytes_per_seq = (size_t)p.pool_h * p.page_size * p.row_bytes;
  half_t* kmax = reinterpret_cast<half_t*>(pg + bytes_per_seq) + (size_t)p.page_size * p.pool_h * 2;
  half_t* kmin = kmax + (size_t)subs * p.pool_h * SDH;
  for (int sc = half_id; sc < subs; sc += 2) {
    const int t0 = page * p.page_size + sc * p.pooling_size;
    if (t0 >= len) break;  // the reference only stores sub-chunks whose first token exists
    half_t mx, mn;
    {
      const half_t x = p.k[((size_t)(begin + t0) * p.num_input_heads + hin) * SDH + d];
      mx = x; mn = x;
    }
    for (int t = 1; t < p.pooling_size; ++t) {
      const int tok = min(t0 + t, len - 1);   // tokens past the end repeat the last one (no effect on min/max)
      const half_t x = p.k[((size_t)(begin + tok) * p.num_input_heads + hin) * SDH + d];
      mx = x > mx ? x : mx;
      mn = x < mn ? x : mn;
    }
    gstore<half_t>(kmax + ((size_t)sc * p.pool_h + r) * SDH + d, mx);
    gstore<half_t>(kmin + ((size_t)sc * p.pool_h + r) * SDH + d, mn);
  }
}

struct SelArgs {
  half_t* out;                // [B, Hq, padded_sub_chunks] (zero filled by the caller)
  const half_t* q; int64_t q_stride;
  const int64_t* kv_pointers; // retrieval table [B,2,max_blocks]
  const int* retrieval_head_flags; const int* head_rank_table; const int* lengths;
  int max_blocks, num_heads, num_kv_heads, num_retrieval_kv_heads, tpb, sub, padded;
  const float* rope; int rope_max_pos;
  int row_bytes;              // Dh/2 (KV4 pages) or Dh (KV8 pages)
};

// Page selector.  One workgroup per (block of SEL_PB pages, kv head, sequence) scores ALL q heads of the kv
// head's GQA group from one pass over the page statistics (the reference walks them once per q head).  A team
// of 16 lanes (= one DPP row; 8 dims per lane, one 16-B load of kmax and of kmin) owns a sub-chunk; a team keeps
// 8 sub-chunks = 16 loads in flight.  Per-lane sums run over e = 0..7 and the team butterfly over xor 8, 4, 2, 1
// exactly as KVPageSelectorTemplate.hpp:482-493, so the fp32 sums (and the fp16 scores) are bit-identical.
constexpr int SEL_PB = 32;     // pages per workgroup
constexpr int SEL_UN = 8;      // sub-chunks in flight per team

__device__ __forceinline__ float team_xor4(float v, int lane) {
  const float up = dpp_mov<0x104>(v);    // row_shl:4 : lane i reads lane i+4
  const float dn = dpp_mov<0x114>(v);    // row_shr:4 : lane i reads lane i-4
  return (lane & 4) ? dn : up;
}

template <int G>
__global__ __launch_bounds__(256) void kv_page_selector_kernel(SelArgs p) {
  __shared__ __attribute__((aligned(16))) half_t q_lds[G * SDH];
  __shared__ int64_t page_lds[SEL_PB];
  const int hk = blockIdx.y, b = blockIdx.z;
  if (p.retrieval_head_flags[hk] == 0) return;   // streaming heads: scores stay zero
  const int rank = p.head_rank_table[hk];
  const int tlen = p.lengths[b] - 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int subs = p.tpb / p.sub;
  const int n_sub = (tlen + p.sub - 1) / p.sub;
  const int page0 = blockIdx.x * SEL_PB;
  const int c0 = page0 * subs;
  if (c0 >= n_sub) return;
  const int h0 = hk * G;
  if (tid < SEL_PB) {
    const int pg = page0 + tid;
    page_lds[tid] = (pg < p.max_blocks && pg * subs < n_sub) ? p.kv_pointers[(size_t)b * 2 * p.max_blocks + pg] : 0;
  }
  for (int idx = tid; idx < G * 64; idx += 256) {   // RoPE(q) at position tlen, rounded to fp16 (as the decode kernel)
    const int g = idx >> 6, i = idx & 63;
    const int rp = tlen < p.rope_max_pos ? tlen : p.rope_max_pos - 1;
    const float* cs = p.rope + (size_t)rp * SDH;
    const half_t* src = p.q + (size_t)b * p.q_stride + (size_t)(h0 + g) * SDH;
    const float c = cs[2 * i], s = cs[2 * i + 1];
    const float a = (float)src[i], bb = (float)src[i + 64];
    const float t0 = c * a, t1 = s * bb, t2 = c * bb, t3 = s * a;
    q_lds[g * SDH + i] = (half_t)(t0 - t1);
    q_lds[g * SDH + i + 64] = (half_t)(t2 + t3);
  }
  __syncthreads();
  const int part = tid & 15, team = tid >> 4;
  v8h q8[G];
#pragma unroll
  for (int g = 0; g < G; ++g) q8[g] = *reinterpret_cast<const v8h*>(q_lds + g * SDH + part * 8);
  const size_t stats_off = (size_t)p.num_retrieval_kv_heads * p.tpb * p.row_bytes +
                           (size_t)p.tpb * p.num_retrieval_kv_heads * 4;                    // bytes: data | scale | zero
  const size_t min_off = (size_t)subs * p.num_retrieval_kv_heads * SDH;                     // halfs: kmax -> kmin
  const int n_here = min(SEL_PB * subs, n_sub - c0);
  for (int i0 = team; i0 < n_here; i0 += 16 * SEL_UN) {
    v8h mx[SEL_UN], mn[SEL_UN];
#pragma unroll
    for (int u = 0; u < SEL_UN; ++u) {   // branch-free: sub-chunks past the end re-read the workgroup's first one
      const int ci = i0 + 16 * u;
      const int cc = ci < n_here ? ci : 0;
      const int pg = cc / subs, sc = cc - pg * subs;
      const half_t* kmax = reinterpret_cast<const half_t*>(reinterpret_cast<const uint8_t*>(page_lds[pg]) + stats_off) +
                           ((size_t)sc * p.num_retrieval_kv_heads + rank) * SDH + part * 8;
      mx[u] = gload<v8h>(kmax);          // global_load (the pointer came out of the int64 page table)
      mn[u] = gload<v8h>(kmax + min_off);
    }
#pragma unroll
    for (int u = 0; u < SEL_UN; ++u) {
      const int ci = i0 + 16 * u;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float acc = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const half_t a = q8[g][e] * mx[u][e], bq = q8[g][e] * mn[u][e];   // fp16 products (hmul2 / hmax2 upstream)
          acc += (float)(a > bq ? a : bq);
        }
        acc += dpp_mov<0x128>(acc);        // xor 8 (row_ror:8)
        acc += team_xor4(acc, lane);       // xor 4
        acc += lane_xor2(acc);
        acc += lane_xor1(acc);
        if (part == 0 && ci < n_here) p.out[((size_t)b * p.num_heads + h0 + g) * p.padded + c0 + ci] = (half_t)acc;
      }
    }
  }
}

}  // namespace omni

using namespace omni;

// ------------------------------------------------------------------------------------------
// Page choice between the selector and the sparse attention (decoding_attention.py:132-142, torch code upstream):
// page score = max over the sub-chunks of a page, then the k best pages of [0, total_pages - 1) plus the newest
// page last.  One workgroup per (sequence, q head): 32-bit keys (orderable fp16 bits << 16 | ~page) in LDS, a
// 4-pass radix select finds the k-th largest key, the survivors are sorted (descending score, as torch.topk
// returns them).  Ties go to the lower page index.  torch needs five kernels (~80 us per layer at 4000 pages).
// ------------------------------------------------------------------------------------------
constexpr int TOPK_MAX_PAGES = 14336;    // dynamic LDS: 56 KiB of keys (+ 8 KiB static: histograms, survivors)
constexpr int TOPK_MAX_K = 1024;

__global__ __launch_bounds__(256) void select_topk_pages_kernel(const half_t* __restrict__ scores, int64_t head_stride,
                                                                int subs, int total_pages, int k,
                                                                int* __restrict__ out) {
  // keys live in dynamic LDS; histograms and survivors are separate static objects so that the compiler may batch the
  // key reads of a loop across the histogram atomics (one LDS round trip per 8 keys instead of one per key)
  extern __shared__ uint32_t keys[];
  __shared__ uint32_t hist[4 * 256];
  __shared__ uint32_t surv[TOPK_MAX_K];
  __shared__ uint32_t s_prefix, s_remaining, s_count, s_done;
  const int n = total_pages - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const half_t* sc = scores + (size_t)blockIdx.x * head_stride;
  int* o = out + (size_t)blockIdx.x * (k + 1);
  auto make_key = [&](half_t m, int p) {
    uint32_t h = __builtin_bit_cast(uint16_t, m);
    if (h == 0x8000u) h = 0;
    const uint32_t k16 = (h & 0x8000u) ? (~h & 0xFFFFu) : (h | 0x8000u);
    return (k16 << 16) | (uint32_t)(0xFFFF - p);
  };
  auto hmax = [](half_t a, half_t b) { return (float)b > (float)a ? b : a; };   // torch.max semantics for finite scores
  if (subs == 4 && (head_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(scores) & 7) == 0) {
    // the launch.sh configuration: one 8-B load per page, eight pages in flight per thread (a loop of dependent
    // 2-B loads costs one memory round trip per page)
    typedef _Float16 v4h_t __attribute__((ext_vector_type(4)));
    for (int p0 = tid; p0 < n; p0 += 256 * 8) {
      v4h_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + 256 * u;
        v[u] = *reinterpret_cast<const v4h_t*>(sc + (size_t)(p < n ? p : p0) * 4);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + 256 * u;
        if (p < n) keys[p] = make_key(hmax(hmax(v[u][0], v[u][1]), hmax(v[u][2], v[u][3])), p);
      }
    }
  } else {
    for (int p = tid; p < n; p += 256) {
      half_t m = sc[(size_t)p * subs];
      for (int j = 1; j < subs; ++j) m = hmax(m, sc[(size_t)p * subs + j]);
      keys[p] = make_key(m, p);
    }
  }
  if (tid == 0) { s_prefix = 0; s_remaining = (uint32_t)k; s_count = 0; s_done = 0; }
  __syncthreads();
  // radix select, most significant byte first: after pass b the k-th largest key starts with s_prefix (b+1 bytes).
  // One histogram per wave (less contention on the few exponent bins), scanned from the top by one wave.
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
#pragma unroll
    for (int w = 0; w < 4; ++w) hist[w * 256 + tid] = 0;
    __syncthreads();
    const uint32_t prefix = s_prefix;
    const uint32_t pmask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int p0 = tid; p0 < n; p0 += 256 * 8) {
      uint32_t kk[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) kk[u] = keys[p0 + 256 * u < n ? p0 + 256 * u : p0];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (p0 + 256 * u < n && (kk[u] & pmask) == prefix) atomicAdd(&hist[wave * 256 + ((kk[u] >> shift) & 0xFFu)], 1u);
    }
    __syncthreads();
    if (wave == 0) {   // lane l owns bins 255-4l .. 252-4l (descending); inclusive scan over the lanes
      uint32_t c[4], tot = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int b = 255 - 4 * lane - j;
        c[j] = hist[b] + hist[256 + b] + hist[512 + b] + hist[768 + b];
        tot += c[j];
      }
      uint32_t incl = tot;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
      }
      const uint32_t rem = s_remaining;
      const unsigned long long reached = __ballot(incl >= rem);
      const int first = __ffsll((long long)reached) - 1;      // rem <= matching keys, so some lane reaches it
      if (lane == first) {
        uint32_t before = incl - tot;
        int j = 0;
        for (; j < 3; ++j) {
          if (before + c[j] >= rem) break;
          before += c[j];
        }
        s_prefix = prefix | ((uint32_t)(255 - 4 * lane - j) << shift);
        s_remaining = rem - before;
        // after the second pass the 16 score bits of the k-th key are known: when EVERY key with that score is taken (the
        // usual case: scores differ) the page bits need no passes -- all keys >= (score << 16) are exactly the k wanted
        if (pass == 1 && rem - before == c[j]) s_done = 1;
      }
    }
    __syncthreads();
    if (s_done) break;      // (workgroup-uniform; s_prefix then has zero page bits)
  }
  const uint32_t kth = s_prefix;      // keys are unique: exactly k keys are >= kth
  for (int p0 = tid; p0 < n; p0 += 256 * 8) {
    uint32_t kk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) kk[u] = keys[p0 + 256 * u < n ? p0 + 256 * u : p0];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (p0 + 256 * u < n && kk[u] >= kth) surv[atomicAdd(&s_count, 1u)] = kk[u];
  }
  __syncthreads();
  // rank sort of the k survivors (k <= 1024: k*k/256 compares per thread)
  for (int i = tid; i < k; i += 256) {
    const uint32_t key = surv[i];
    int rank = 0;
    int j = 0;
    for (; j + 8 <= k; j += 8) {
      uint32_t t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = surv[j + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) rank += t[u] > key ? 1 : 0;
    }
    for (; j < k; ++j) rank += surv[j] > key ? 1 : 0;
    o[rank] = 0xFFFF - (int)(key & 0xFFFFu);
  }
  if (tid == 0) o[k] = total_pages - 1;
}

extern "C" int omni_select_topk_pages(void* out_i32, const void* scores_f16, int64_t head_stride, int heads_total,
                                      int subs_per_page, int total_pages, int k, void* stream) {
  if (!out_i32 || !scores_f16 || heads_total < 0 || subs_per_page < 1 || total_pages < 1 || k < 0) return OMNI_EINVAL;
  if (k > total_pages - 1 || k > TOPK_MAX_K || total_pages - 1 > TOPK_MAX_PAGES ||
      head_stride < (int64_t)(total_pages - 1) * subs_per_page)
    return OMNI_EINVAL;
  if (heads_total == 0) return OMNI_OK;
  const size_t lds = (size_t)(total_pages - 1) * sizeof(uint32_t);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)select_topk_pages_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(select_topk_pages_kernel, dim3(heads_total), dim3(256), lds, (hipStream_t)stream,
                     (const half_t*)scores_f16, head_stride, subs_per_page, total_pages, k, (int*)out_i32);
  return omni_launch_status();
}

extern "C" int omni_kv_min_max_pool(const void* k_f16, const void* kv_pointers_i64, const void* cu_seqlens_i32,
                                    const void* pooling_heads_idx_i32, int batch, int max_blocks, int num_input_heads,
                                    int num_pool_heads, int head_dim, int kv_row_bytes, int max_seqlen, int pooling_size,
                                    int page_size, void* stream) {
  if (!k_f16 || !kv_pointers_i64 || !cu_seqlens_i32 || !pooling_heads_idx_i32) return OMNI_EINVAL;
  if (head_dim != SDH || batch < 1 || num_pool_heads < 0 || pooling_size < 1 || page_size % pooling_size != 0 ||
      max_seqlen < 0 || (kv_row_bytes != SDH / 2 && kv_row_bytes != SDH))
    return OMNI_EINVAL;
  if (num_pool_heads == 0 || max_seqlen == 0) return OMNI_OK;
  PoolArgs a{(const half_t*)k_f16, (const int64_t*)kv_pointers_i64, (const int*)cu_seqlens_i32,
             (const int*)pooling_heads_idx_i32, max_blocks, num_input_heads, num_pool_heads, pooling_size, page_size,
             kv_row_bytes};
  dim3 grid((max_seqlen + page_size - 1) / page_size, batch, num_pool_heads);
  hipLaunchKernelGGL(kv_min_max_pool_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  return omni_launch_status();
}

extern "C" int omni_kv_page_selector(void* out_f16, const void* q_f16, int64_t q_stride, const void* kv_pointers_i64,
                                     const void* retrieval_head_flags_i32, const void* head_rank_table_i32,
                                     const void* lengths_i32, int batch, int max_blocks, int num_heads, int num_kv_heads,
                                     int num_retrieval_kv_heads, int head_dim, int kv_row_bytes, int tokens_per_block,
                                     int tokens_per_sub_chunk, int padded_sub_chunks, const void* rope_cos_sin_f32,
                                     int rope_max_pos, void* stream) {
  if (!out_f16 || !q_f16 || !kv_pointers_i64 || !retrieval_head_flags_i32 || !head_rank_table_i32 || !lengths_i32 ||
      !rope_cos_sin_f32)
    return OMNI_EINVAL;
  if (head_dim != SDH || batch < 1 || num_heads < 1 || num_kv_heads < 1 || num_heads % num_kv_heads != 0 ||
      tokens_per_sub_chunk < 1 || tokens_per_block % tokens_per_sub_chunk != 0 || padded_sub_chunks < 0 ||
      (kv_row_bytes != SDH / 2 && kv_row_bytes != SDH))
    return OMNI_EINVAL;
  SelArgs a{(half_t*)out_f16, (const half_t*)q_f16, q_stride, (const int64_t*)kv_pointers_i64,
            (const int*)retrieval_head_flags_i32, (const int*)head_rank_table_i32, (const int*)lengths_i32,
            max_blocks, num_heads, num_kv_heads, num_retrieval_kv_heads, tokens_per_block, tokens_per_sub_chunk,
            padded_sub_chunks, (const float*)rope_cos_sin_f32, rope_max_pos, kv_row_bytes};
  const int group = num_heads / num_kv_heads;
  const int subs = tokens_per_block / tokens_per_sub_chunk;
  const int pages = (padded_sub_chunks + subs - 1) / subs;
  if (pages == 0) return OMNI_OK;
  dim3 grid((pages + SEL_PB - 1) / SEL_PB, num_kv_heads, batch);
  switch (group) {
    case 1: hipLaunchKernelGGL(kv_page_selector_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
    case 2: hipLaunchKernelGGL(kv_page_selector_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
    case 4: hipLaunchKernelGGL(kv_page_selector_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
    case 8: hipLaunchKernelGGL(kv_page_selector_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
    default: return OMNI_EINVAL;
  }
  return omni_launch_status();
}
