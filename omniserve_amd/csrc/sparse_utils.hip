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
  const size_t bytes_per_seq = (size_t)p.pool_h * p.page_size * (SDH / 2);
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
    kmax[((size_t)sc * p.pool_h + r) * SDH + d] = mx;
    kmin[((size_t)sc * p.pool_h + r) * SDH + d] = mn;
  }
}

struct SelArgs {
  half_t* out;                // [B, Hq, padded_sub_chunks] (zero filled by the caller)
  const half_t* q; int64_t q_stride;
  const int64_t* kv_pointers; // retrieval table [B,2,max_blocks]
  const int* retrieval_head_flags; const int* head_rank_table; const int* lengths;
  int max_blocks, num_heads, num_kv_heads, num_retrieval_kv_heads, tpb, sub, padded;
  const float* rope; int rope_max_pos;
};

// one workgroup per (q head, sequence); 16 lanes per sub-chunk (8 dims each)
__global__ __launch_bounds__(256) void kv_page_selector_kernel(SelArgs p) {
  __shared__ __attribute__((aligned(16))) half_t q_lds[SDH];
  const int h = blockIdx.x, b = blockIdx.y;
  const int hk = h / (p.num_heads / p.num_kv_heads);
  if (p.retrieval_head_flags[hk] == 0) return;   // streaming heads: scores stay zero
  const int rank = p.head_rank_table[hk];
  const int tlen = p.lengths[b] - 1;
  const int tid = threadIdx.x;
  if (tid < 64) {  // RoPE(q) at position tlen, rounded to fp16 (as the decode kernel)
    const int rp = tlen < p.rope_max_pos ? tlen : p.rope_max_pos - 1;
    const float* cs = p.rope + (size_t)rp * SDH;
    const half_t* src = p.q + (size_t)b * p.q_stride + (size_t)h * SDH;
    const float c = cs[2 * tid], s = cs[2 * tid + 1];
    const float a = (float)src[tid], bb = (float)src[tid + 64];
    const float t0 = c * a, t1 = s * bb, t2 = c * bb, t3 = s * a;
    q_lds[tid] = (half_t)(t0 - t1);
    q_lds[tid + 64] = (half_t)(t2 + t3);
  }
  __syncthreads();
  const int n_sub = (tlen + p.sub - 1) / p.sub;
  const int subs = p.tpb / p.sub;
  const int part = tid & 15;                 // 8 dims
  const v8h q8 = *reinterpret_cast<const v8h*>(q_lds + part * 8);
  const size_t bytes_per_seq = (size_t)p.num_retrieval_kv_heads * p.tpb * (SDH / 2);
  for (int c = tid >> 4; c < n_sub; c += 16) {
    const int page = (c * p.sub) / p.tpb, sc = c % subs;
    const uint8_t* pg = reinterpret_cast<const uint8_t*>(p.kv_pointers[(size_t)b * 2 * p.max_blocks + page]);
    const half_t* kmax = reinterpret_cast<const half_t*>(pg + bytes_per_seq) + (size_t)p.tpb * p.num_retrieval_kv_heads * 2;
    const half_t* kmin = kmax + (size_t)subs * p.num_retrieval_kv_heads * SDH;
    const size_t o = ((size_t)sc * p.num_retrieval_kv_heads + rank) * SDH + part * 8;
    const v8h mx = *reinterpret_cast<const v8h*>(kmax + o);
    const v8h mn = *reinterpret_cast<const v8h*>(kmin + o);
    float acc = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const half_t a = q8[e] * mx[e], bq = q8[e] * mn[e];   // fp16 products, as the reference (hmul2 / hmax2)
      acc += (float)(a > bq ? a : bq);
    }
#pragma unroll
    for (int m = 8; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (part == 0) p.out[((size_t)b * p.num_heads + h) * p.padded + c] = (half_t)acc;
  }
}

}  // namespace omni

using namespace omni;

extern "C" int omni_kv_min_max_pool(const void* k_f16, const void* kv_pointers_i64, const void* cu_seqlens_i32,
                                    const void* pooling_heads_idx_i32, int batch, int max_blocks, int num_input_heads,
                                    int num_pool_heads, int head_dim, int max_seqlen, int pooling_size, int page_size,
                                    void* stream) {
  if (!k_f16 || !kv_pointers_i64 || !cu_seqlens_i32 || !pooling_heads_idx_i32) return OMNI_EINVAL;
  if (head_dim != SDH || batch < 1 || num_pool_heads < 0 || pooling_size < 1 || page_size % pooling_size != 0 ||
      max_seqlen < 0)
    return OMNI_EINVAL;
  if (num_pool_heads == 0 || max_seqlen == 0) return OMNI_OK;
  PoolArgs a{(const half_t*)k_f16, (const int64_t*)kv_pointers_i64, (const int*)cu_seqlens_i32,
             (const int*)pooling_heads_idx_i32, max_blocks, num_input_heads, num_pool_heads, pooling_size, page_size};
  dim3 grid((max_seqlen + page_size - 1) / page_size, batch, num_pool_heads);
  hipLaunchKernelGGL(kv_min_max_pool_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  return omni_launch_status();
}

extern "C" int omni_kv_page_selector(void* out_f16, const void* q_f16, int64_t q_stride, const void* kv_pointers_i64,
                                     const void* retrieval_head_flags_i32, const void* head_rank_table_i32,
                                     const void* lengths_i32, int batch, int max_blocks, int num_heads, int num_kv_heads,
                                     int num_retrieval_kv_heads, int head_dim, int tokens_per_block,
                                     int tokens_per_sub_chunk, int padded_sub_chunks, const void* rope_cos_sin_f32,
                                     int rope_max_pos, void* stream) {
  if (!out_f16 || !q_f16 || !kv_pointers_i64 || !retrieval_head_flags_i32 || !head_rank_table_i32 || !lengths_i32 ||
      !rope_cos_sin_f32)
    return OMNI_EINVAL;
  if (head_dim != SDH || batch < 1 || num_heads < 1 || num_kv_heads < 1 || num_heads % num_kv_heads != 0 ||
      tokens_per_sub_chunk < 1 || tokens_per_block % tokens_per_sub_chunk != 0 || padded_sub_chunks < 0)
    return OMNI_EINVAL;
  SelArgs a{(half_t*)out_f16, (const half_t*)q_f16, q_stride, (const int64_t*)kv_pointers_i64,
            (const int*)retrieval_head_flags_i32, (const int*)head_rank_table_i32, (const int*)lengths_i32,
            max_blocks, num_heads, num_kv_heads, num_retrieval_kv_heads, tokens_per_block, tokens_per_sub_chunk,
            padded_sub_chunks, (const float*)rope_cos_sin_f32, rope_max_pos};
  hipLaunchKernelGGL(kv_page_selector_kernel, dim3(num_heads, batch), dim3(256), 0, (hipStream_t)stream, a);
  return omni_launch_status();
}
