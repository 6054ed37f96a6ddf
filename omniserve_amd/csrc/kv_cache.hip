// KV4 paged-cache kernels for MI355X (gfx950): padding offsets, prefill writer, decode attention.
//
// Replaces omniserve_backend.fused_attention_pure_dense.single_query_attention,
// omniserve_backend.fused_attention_fine_grained_dense.apply_bias_rope_update_kv_cache (dense
// heads) and compute_padding_offsets (reference: kernels/csrc/fused_attention/...).
//
// Page (per layer, K or V):  int4 data [H_kv][tpb][Dh/2] | fp16 scale [H_kv][tpb] | fp16 zero [H_kv][tpb]
// (per-tensor KV8 pages: int8 data [H][tpb][Dh] and the same 4 B/token-head tail).
// A token row of one head is Dh/2 = 64 contiguous bytes, so 16 consecutive tokens of a head are
// 1 KiB contiguous: one 16-B load per lane per wave.  Design (not a port of the TRT-LLM MMHA):
//   * one workgroup per (kv head, sequence, KV split) serves ALL q heads of the GQA group, so the
//     packed K/V bytes are fetched and dequantised once instead of once per q head;
//   * flash-decoding inside the workgroup on the matrix cores (see the decode kernel below); partial
//     (max, sum, O) per split are merged by a second tiny kernel (or by the quantiser that follows), so
//     B*H_kv*S workgroups fill all 256 CUs;
//   * dequant is the reference's fp16 fma(u4, scale, -scale*zero); dot products and P.V
//     accumulate in fp32.
#include "common.h"
#include "row_kernels.h"

namespace omni {

constexpr int DH = 128;          // head dim (the only one the QServe/LServe models use)
constexpr int ROW_BYTES = DH / 2;

struct KvLayout {
  int tpb;            // tokens per block (power of two, multiple of 16)
  int tpb_log2;
  int num_kv_heads;
  int bytes_per_seq;  // H_kv * tpb * Dh/2  (offset of the scale tail)
};

__host__ __device__ inline KvLayout make_layout(int tpb, int hkv) {
  KvLayout l;
  l.tpb = tpb;
  int lg = 0;
  while ((1 << lg) < tpb) ++lg;
  l.tpb_log2 = lg;
  l.num_kv_heads = hkv;
  l.bytes_per_seq = hkv * tpb * ROW_BYTES;
  return l;
}

// LServe fine-grained head classes (fused_attention_fine_grained/...): every kv head is either a
// retrieval head (full history in the retrieval pool) or a streaming head (sink + local window in a
// ring of pages in the streaming pool); head_rank = index of the head inside its pool's pages.
struct FgArgs {
  const int64_t* strm_pointers;  // [B,2,strm_blocks]
  const int* flags;              // [Hkv] != 0: retrieval head      (nullptr: every head is a dense head)
  const int* rank;               // [Hkv]
  const int* dyn;                // [B,Hq,num_dyn] selected retrieval pages (nullptr: all pages)
  int strm_blocks, num_dyn;
  int num_retr, num_strm;        // heads per page of the two pools
  int sink, local, sink_blocks, local_blocks;
  int sub_chunk;                 // > 0: K pages carry min/max statistics, updated on append
};

__device__ __forceinline__ int ring_block(int blk, int sink_blocks, int local_blocks) {
  return blk < sink_blocks ? blk : sink_blocks + (blk - sink_blocks) % local_blocks;
}

// ------------------------------------------------------------------------------------------
// compute_padding_offsets  (common/input_metadata_helper.cu:16-50)
// ------------------------------------------------------------------------------------------
__global__ void padding_offsets_kernel(int* __restrict__ out, const int* __restrict__ cu, int max_len) {
  const int b = blockIdx.x;
  const int begin = cu[b], end = cu[b + 1];
  const int off = b * max_len - begin;
  for (int i = begin + threadIdx.x; i < end; i += blockDim.x) out[i] = off;
}

// ------------------------------------------------------------------------------------------
// KV4 quantisation helpers (pure_dense/decoderMaskedMultiheadAttentionUtils.h:1838-1884,2070-2077)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t kv4_code(float x, float inv_scale, float zero) {
  float v = x * inv_scale;
  v = v + zero;
  return rni_sat_u8(v) & 0xFu;
}

// ------------------------------------------------------------------------------------------
// Prefill writer.  16 lanes own one head row: lane l holds elements [4l,4l+4) and [64+4l,64+4l+4)
// (the two halves of 4 neox RoPE pairs).  A workgroup of 256 threads handles 16 head slots;
// slots per token = Hq (q heads: RoPE in place) + Hkv (k: RoPE in place + quantise; v: quantise).
// ------------------------------------------------------------------------------------------
// KV8: the per-tensor int8 format of fused_attention_per_tensor (static scales kv_oq[0] for K, kv_oq[1] for V:
// per_tensor_common/applyBiasRopeUpdateKVCache.h:329-338,508-511); a token row of one head is Dh bytes.
template <bool KV8>
__global__ __launch_bounds__(256) void kv4_prefill_write_kernel(
    half_t* __restrict__ qkv, const int* __restrict__ seq_lens, const int* __restrict__ padding_offsets,
    const int64_t* __restrict__ kv_pointers, int tokens, int max_blocks, int num_heads, int num_kv_heads,
    int max_seq_len, KvLayout lay, const float* __restrict__ rope, int rope_max_pos, int cyclic_len, FgArgs fg,
    const float* __restrict__ kv_oq) {
  constexpr int RB = KV8 ? DH : ROW_BYTES;
  const int slots_per_token = num_heads + num_kv_heads;
  const long long slot_id = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l = threadIdx.x & 15;
  if (slot_id >= (long long)tokens * slots_per_token) return;
  const int tok = (int)(slot_id / slots_per_token);
  const int hs = (int)(slot_id % slots_per_token);
  const int g = tok + padding_offsets[tok];
  const int b = g / max_seq_len;
  const int pos = g % max_seq_len;
  const int actual_len = seq_lens[b];
  if (pos >= actual_len) return;  // padding row (cannot happen with consistent metadata)
  const int row_elems = (num_heads + 2 * num_kv_heads) * DH;
  half_t* row = qkv + (size_t)tok * row_elems;
  const bool is_kv = hs >= num_heads;
  const int hk = hs - num_heads;
  half_t* x = is_kv ? row + (size_t)(num_heads + hk) * DH : row + (size_t)hs * DH;

  // RoPE (neox): pair (i, i+64), coefficients from the host table [pos][64][2]
  const int rp = pos < rope_max_pos ? pos : rope_max_pos - 1;
  const float* cs = rope + ((size_t)rp * (DH / 2) + 4 * l) * 2;
  const v4f cs0 = *reinterpret_cast<const v4f*>(cs);
  const v4f cs1 = *reinterpret_cast<const v4f*>(cs + 4);
  const float c[4] = {cs0[0], cs0[2], cs1[0], cs1[2]};
  const float s[4] = {cs0[1], cs0[3], cs1[1], cs1[3]};
  typedef _Float16 v4h __attribute__((ext_vector_type(4)));
  const v4h lo = *reinterpret_cast<const v4h*>(x + 4 * l);
  const v4h hi = *reinterpret_cast<const v4h*>(x + 64 + 4 * l);
  v4h rlo, rhi;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = (float)lo[j], bb = (float)hi[j];
    const float t0 = c[j] * a, t1 = s[j] * bb;
    const float t2 = c[j] * bb, t3 = s[j] * a;
    rlo[j] = (half_t)(t0 - t1);
    rhi[j] = (half_t)(t2 + t3);
  }
  *reinterpret_cast<v4h*>(x + 4 * l) = rlo;
  *reinterpret_cast<v4h*>(x + 64 + 4 * l) = rhi;
  if (!is_kv) return;

  // which tokens a head stores (applyBiasRopeUpdateKVCache.h:296-311): retrieval / dense heads the last
  // cyclic_len tokens (all of them in practice), streaming heads the sinks and the local window
  int page = pos >> lay.tpb_log2;
  const int slot = pos & (lay.tpb - 1);
  const int64_t* tab = kv_pointers + (size_t)b * 2 * max_blocks;
  int tab_blocks = max_blocks, hrank = hk, hpool = lay.num_kv_heads;
  bool streaming = false;
  if (fg.flags) {
    hrank = fg.rank[hk];
    streaming = fg.flags[hk] == 0;
    hpool = streaming ? fg.num_strm : fg.num_retr;
  }
  if (streaming) {
    if (!(pos < fg.sink || pos >= actual_len - fg.local)) return;
    page = ring_block(page, fg.sink_blocks, fg.local_blocks);
    tab = fg.strm_pointers + (size_t)b * 2 * fg.strm_blocks;
    tab_blocks = fg.strm_blocks;
  } else {
    int lower = actual_len - cyclic_len;
    if (lower < 0) lower = 0;
    if (pos < lower) return;
  }
  const int pool_bytes_per_seq = hpool * lay.tpb * RB;
  const half_t* v = row + (size_t)(num_heads + num_kv_heads + hk) * DH;
  const v4h vlo = *reinterpret_cast<const v4h*>(v + 4 * l);
  const v4h vhi = *reinterpret_cast<const v4h*>(v + 64 + 4 * l);

#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const v4h a = which == 0 ? rlo : vlo;
    const v4h bq = which == 0 ? rhi : vhi;
    if constexpr (KV8) {
      const float oq = kv_oq[which];
      float amax = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf((float)a[j]), __builtin_fabsf((float)bq[j])));
#pragma unroll
      for (int m = 8; m > 0; m >>= 1) amax = __builtin_fmaxf(amax, __shfl_xor(amax, m, 64));
      uint8_t* base = reinterpret_cast<uint8_t*>(tab[which * tab_blocks + page]);
      uint8_t* dst = base + ((size_t)hrank * lay.tpb + slot) * RB;
      uint32_t w0 = 0, w1 = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        w0 |= ((uint32_t)rni_sat_s8(oq * (float)a[j]) & 0xFFu) << (8 * j);
        w1 |= ((uint32_t)rni_sat_s8(oq * (float)bq[j]) & 0xFFu) << (8 * j);
      }
      *reinterpret_cast<uint32_t*>(dst + 4 * l) = w0;
      *reinterpret_cast<uint32_t*>(dst + 64 + 4 * l) = w1;
      // the row's own h(absmax/127) still goes to the scale slot of the tail (:387-414); nothing reads it
      if (l == 0)
        (reinterpret_cast<half_t*>(base + pool_bytes_per_seq) + hrank * lay.tpb + slot)[0] = (half_t)(amax / 127.0f);
      continue;
    }
    float mx = (float)a[0], mn = (float)a[0];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mx = __builtin_fmaxf(mx, __builtin_fmaxf((float)a[j], (float)bq[j]));
      mn = __builtin_fminf(mn, __builtin_fminf((float)a[j], (float)bq[j]));
    }
#pragma unroll
    for (int m = 8; m > 0; m >>= 1) {
      mx = __builtin_fmaxf(mx, __shfl_xor(mx, m, 64));
      mn = __builtin_fminf(mn, __shfl_xor(mn, m, 64));
    }
    const float range = mx - mn;
    const half_t scale_h = (half_t)(range / 15.0f);
    const float nm = -15.0f * mn;
    const half_t zero_h = (half_t)(nm / range);
    const float inv = 1.0f / (float)scale_h;
    const float z = (float)zero_h;
    uint8_t* base = reinterpret_cast<uint8_t*>(tab[which * tab_blocks + page]);
    uint8_t* dst = base + ((size_t)hrank * lay.tpb + slot) * ROW_BYTES;
    uint32_t q[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      q[j] = kv4_code((float)a[j], inv, z);
      q[4 + j] = kv4_code((float)bq[j], inv, z);
    }
    *reinterpret_cast<uint16_t*>(dst + 2 * l) = (uint16_t)(q[0] | (q[1] << 4) | (q[2] << 8) | (q[3] << 12));
    *reinterpret_cast<uint16_t*>(dst + 32 + 2 * l) = (uint16_t)(q[4] | (q[5] << 4) | (q[6] << 8) | (q[7] << 12));
    if (l == 0) {
      half_t* sc = reinterpret_cast<half_t*>(base + pool_bytes_per_seq) + hrank * lay.tpb + slot;
      sc[0] = scale_h;
      sc[hpool * lay.tpb] = zero_h;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Decode attention
// ------------------------------------------------------------------------------------------
// unpack 16 packed bytes (32 codes) and dequantise with fp16 fma(u, scale, c), c = h(-scale*zero).
// Output order per 32-bit word of 8 codes e0..e7: half2 pairs (e0,e4) (e1,e5) (e2,e6) (e3,e7),
// i.e. out[w*4 + j] = {e_j, e_{j+4}} of word w (the q operand is loaded in the same order).
// (x & mask) | magic in one VALU op.  VOP3 on gfx9 takes one scalar operand, so the magic lives in a VGPR; the
// compiler emits v_and + v_or with two literals otherwise (64 extra ops per 32-token tile of the decode kernels).
__device__ __forceinline__ uint32_t and_or(uint32_t x, uint32_t mask, uint32_t magic) {
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(mask), "v"(magic));
  return r;
}

template <bool CHEAP = false>
__device__ __forceinline__ void kv4_dequant16(const uint4 raw, v2h scale2, v2h c2, v2h out[16]) {
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
  if constexpr (CHEAP) {      // (timing experiments only: 16 instead of 52 operations, no dependent chains)
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = __builtin_bit_cast(v2h, w[i & 3]) * scale2;
    return;
  }
  const v2h k1024 = {(half_t)1024.0f, (half_t)1024.0f};
  const v2h k16th = {(half_t)0.0625f, (half_t)0.0625f};
  const v2h k64 = {(half_t)64.0f, (half_t)64.0f};
  const uint32_t magic = 0x64006400u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t x = w[i], y = w[i] >> 8;
    uint32_t h0 = and_or(x, 0x000f000fu, magic);  // {1024+e0, 1024+e4}
    uint32_t h1 = and_or(x, 0x00f000f0u, magic);  // {1024+16 e1, 1024+16 e5}
    uint32_t h2 = and_or(y, 0x000f000fu, magic);  // e2, e6
    uint32_t h3 = and_or(y, 0x00f000f0u, magic);  // e3, e7
    v2h u0 = __builtin_bit_cast(v2h, h0) - k1024;
    v2h u1 = __builtin_elementwise_fma(__builtin_bit_cast(v2h, h1), k16th, -k64);
    v2h u2 = __builtin_bit_cast(v2h, h2) - k1024;
    v2h u3 = __builtin_elementwise_fma(__builtin_bit_cast(v2h, h3), k16th, -k64);
    out[i * 4 + 0] = __builtin_elementwise_fma(u0, scale2, c2);
    out[i * 4 + 1] = __builtin_elementwise_fma(u1, scale2, c2);
    out[i * 4 + 2] = __builtin_elementwise_fma(u2, scale2, c2);
    out[i * 4 + 3] = __builtin_elementwise_fma(u3, scale2, c2);
  }
}

// KV8 (per-tensor int8): out[i] = {h(s*f32(b_2i)), h(s*f32(b_2i+1))} for the 16 bytes of `raw`, natural order
// (common/decoderMaskedMultiheadAttentionUtils.h:2086-2093: f32 multiply, then one rounding to fp16)
__device__ __forceinline__ void kv8_dequant16(const uint4 raw, float s, v2h out[8]) {
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int x = (int)w[i];
    const float f0 = (float)((x << 24) >> 24) * s, f1 = (float)((x << 16) >> 24) * s;
    const float f2 = (float)((x << 8) >> 24) * s, f3 = (float)(x >> 24) * s;
    out[2 * i] = (v2h){(half_t)f0, (half_t)f1};
    out[2 * i + 1] = (v2h){(half_t)f2, (half_t)f3};
  }
}

// Fused extension: the q / k / v rows of the current token as the qkv projection's int32 split-K slabs (the projection's
// GEMM epilogue -- qgemm_kernel.h epilogue<> -- is applied by the attention kernel's first load trip instead of by a slab
// epilogue launch between the two).  slab == nullptr: plain fp16 q / k / v.
struct QkvSlabSrc {
  const int32_t* slab;     // [sk][M][N] int32 partial sums of the qkv projection (omni_*_gemm_partial)
  long long sstride;       // M * N
  int sk, n;               // slabs, row width N
  int col_q, col_k, col_v; // first output channel of the q / k / v blocks inside a row
  const half_t* wscales;   // [N]
  const half_t* wsz;       // [N] per-channel W4A8 zero term, or nullptr (W8A8 / per-group epilogue)
  const half_t* ascales;   // [M] scales (and sums) of the projection's int8 input
  const half_t* asum;      // nullptr with wsz
};

struct DecodeArgs {
  half_t* out;             // [B,Hq,128]
  const half_t* q;         // row stride q_stride
  const half_t* k;
  const half_t* v;
  int64_t q_stride, kv_stride;
  const int64_t* kv_pointers;  // [B,2,max_blocks]
  const int* lengths;
  int batch, max_blocks, num_heads, num_kv_heads;
  KvLayout lay;
  int nsplit;              // KV splits (grid.x)
  int split_tokens;        // max tokens per split (multiple of 16)
  const float* rope;
  int rope_max_pos;
  float* part_ml;          // [B,Hq,S,2]
  float* part_o;           // [B,Hq,S,128]
  FgArgs fg;               // fine-grained (retrieval / streaming) extension, used by the FG instantiations
  const float* kv_qo;      // KV8 instantiations: device fp32 [2] kv_scale_quant_orig (K, V) ...
  const float* kv_oq;      // ... and kv_scale_orig_quant
  QkvSlabSrc qs;           // q / k / v from the projection's slabs (slab != nullptr)
  // single-launch form (LASTM instantiation, fused extension): the workgroup that arrives LAST at its (sequence, head group)
  // ticket merges the splits' partials, writes the fp16 output and raises the row maxima -- no merge launch
  uint32_t* tickets;       // [B * gridDim.y] zero between launches (the last arriver resets its word)
  uint32_t* amax;          // row-maximum slots (common.h: amax_raise) of the fp16 output
  PrefetchArgs pf;         // armed L2 prefetch riding on extra z slices of the grid (pf.first_block = their first linear id)
};

constexpr int DEC_MAX_SPLITS = 1024;   // KV splits per (sequence, head group): contexts up to 1024 x 2048 tokens
constexpr int DEC_THREADS = 256;
constexpr int DEC_WAVES = DEC_THREADS / 64;

// position of head-dim element d in the "dequant order" used for q in LDS: within every 8-dim
// word the pairs (j, j+4) are adjacent, matching the half2 pairs kv4_dequant16 produces.
__device__ __forceinline__ int perm_pos(int d) {
  const int r = d & 7;
  return (d & ~7) | ((r & 3) << 1) | (r >> 2);
}

// ------------------------------------------------------------------------------------------
// Decode attention on the matrix cores (v_mfma_f32_16x16x32_f16), flash-decoding inside the workgroup:
//   * Q.K^T: A = dequantised K (row = token, k = head dims: one packed dword = 8 codes = one
//     lane's 8 k-slots), B = q (col = q head, padded to 16), 4 MFMAs per 16 tokens;
//   * P.V: V is stored token-major, the contraction runs over tokens, so the dequantised fp16 V
//     tile (32 tokens x 128 dims) goes through LDS once and comes back transposed with
//     ds_read_b64_tr_b16 (lane L of a 16-lane group passes &M[L>>2][4*(L&3)] and receives column
//     L of the 4x16 block -- verified on hardware with tools/tr_probe.hip); A = V^T (row = dim),
//     B = P^T (col = q head, probabilities rounded to fp16 as the reference does), 8 MFMAs per
//     32 tokens; the O accumulators are 32 VGPRs.
// (Two earlier versions -- a VALU kernel with fdot2 and a two-pass MFMA kernel that kept all scores of
// a split in LDS -- were 15-40 % slower and are gone; see DESIGN.md 4.4 and profiles/r01_c, r01_e.)
// ------------------------------------------------------------------------------------------
typedef __fp16 v4hp __attribute__((__vector_size__(4 * sizeof(__fp16))));

__device__ __forceinline__ int unperm_pos(int q) { return (q & ~7) | ((q >> 1) & 3) | ((q & 1) << 2); }

// One sweep over the split: Q.K^T -> online softmax in registers -> P.V per 32-token tile.
// FG: LServe fine-grained mode.  The split runs over a head's list of *attended* cached tokens
// ("virtual" tokens): all of them for a retrieval head, the tokens of the selected pages with
// fg.dyn (one list per q head, hence G = 1), min(sink+local-1, tlen) tokens read through the page
// ring for a streaming head (decoderMaskedMultiheadAttentionTemplate.hpp:1475-1537 of
// fused_attention_fine_grained/dense_attention, :1566-1641 of sparse_attention).
constexpr int FVROW = 288;          // V tile row pitch of the flash kernel (see the layout note in the kernel)
constexpr int FVTILE = 32 * FVROW;
#ifndef OMNI_FLASH_MIN_BLOCKS
#define OMNI_FLASH_MIN_BLOCKS 2
#endif
#ifndef OMNI_FLASH_FB
#define OMNI_FLASH_FB 1        // 32-token tiles per wave whose K and V bytes are requested together.  1: 148 VGPRs (three
                               // workgroups per CU by registers); 2: 197 VGPRs, two per CU.  Isolated launches time the same
                               // (15 / 25 / 88 us at 16 x 1 K, 64 x 1 K, 8 x 32 K); inside the decode step 1 is 1-2 % faster per
                               // STEP at bs = 16 and bs = 64 (2.31 -> 2.26-2.29 ms, 3.59 -> 3.53-3.54 ms, profiles/r03_h)
#endif
#ifndef OMNI_FLASH_ABLATE
#define OMNI_FLASH_ABLATE 0    // timing experiments (WRONG results), bits: 1 loads only, 2 no P.V part (V unpack, LDS round trip, MFMAs),
#endif                         // 4 V unpack without its arithmetic, 8 K unpack without its arithmetic, 16 no exponentials (profiles/r05_b)
#ifndef OMNI_FLASH_SLOTS
#define OMNI_FLASH_SLOTS 512   // workgroups the chip holds at a time (256 CUs x workgroups per CU): the split planner's round size
#endif
// KV8: per-tensor int8 pages (fused_attention_per_tensor): rows of Dh bytes, dequant h(kv_qo * f32(int8)) with the static
// scales kv_qo[0] (K) / kv_qo[1] (V), natural element order (q is not reordered), append with kv_oq, no tail write.
// LASTM (dense KV4, splits > 1): partials are written through, every split workgroup takes a ticket of its (sequence, head
// group), and the LAST arriver merges the splits (row_kernels.h: SrcAttnMerge, the merge kernels' own arithmetic), stores
// the fp16 output and raises its row maxima: the attention is ONE launch (it also carries the armed L2 prefetch on extra
// z slices of its grid, as the merge kernel did).
#if (OMNI_FLASH_ABLATE & 64)
// timing experiment (tools/attn_long.py with OMNI_DBG_POOL=1; results are wrong unless the pools are laid out in table order): trip 1
// without its loads -- page pointers from arithmetic on a pool base and the length from a constant, both read through the scalar
// cache -- i.e. what a persistent workgroup that fetched the next item's length / page window during its current sweep would see
static __device__ unsigned long long omni_dbg_pool[4];      // K pool base, V pool base, page bytes, cached tokens per sequence
extern "C" int omni_debug_set_pool(unsigned long long kbase, unsigned long long vbase, unsigned long long page_bytes, unsigned long long tlen) {
  const unsigned long long h[4] = {kbase, vbase, page_bytes, tlen};
  return hipMemcpyToSymbol(HIP_SYMBOL(omni_dbg_pool), h, sizeof(h)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef OMNI_DEBUG_CLOCKS
// timeline probe (tools/flash_timeline.py): shader-clock stamps of every wave of three workgroups -- the first, the middle
// and the last of the grid in dispatch order: [0] entry, [1] page window visible, [2] q in LDS (first batch requested),
// [3 + i] tile i of the wave done (i < 22), [25] partials stored (single-launch form), [26] stores drained, [27] ticket known,
// [28] sweep done, [29] combined / stored (single-launch form: merged by the last arriver), [30] exit
static __device__ unsigned long long omni_dbg_flash[3 * DEC_WAVES * 32];
#define FLASH_STAMP(i)                                                                                              \
  do {                                                                                                              \
    if (fdbg >= 0 && (threadIdx.x & 63) == 0) omni_dbg_flash[(fdbg * DEC_WAVES + (threadIdx.x >> 6)) * 32 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define FLASH_STAMP(i) do {} while (0)
#endif
// PIPE (dense instantiations; the host picks it when a wave sweeps six or more tiles): requests two tiles ahead, see below.
template <int G, bool DIRECT, bool FG = false, bool KV8 = false, bool LASTM = false, bool PIPE = true>
__global__ __launch_bounds__(DEC_THREADS, OMNI_FLASH_MIN_BLOCKS) void kv4_decode_flash_kernel(DecodeArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
#ifdef OMNI_DEBUG_CLOCKS
  const unsigned flin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), fall = gridDim.x * gridDim.y * gridDim.z;
  const int fdbg = flin == 0 ? 0 : (flin == fall / 2 ? 1 : (flin == fall - 1 ? 2 : -1));
  FLASH_STAMP(0);
#endif
  // One batch of scalar loads for the kernel arguments of trip 1 (and the rider test): hipcc otherwise requests them where
  // they are first used -- three dependent kernarg round trips (rider test, geometry, pointers) in front of the first vector
  // load of a kernel that is a chain of memory round trips (ISA of round 4, profiles/r04_e).
  asm volatile("" ::"s"(p.out), "s"(p.q), "s"(p.k), "s"(p.v), "s"(p.q_stride), "s"(p.kv_stride), "s"(p.kv_pointers), "s"(p.lengths),
               "s"(p.batch), "s"(p.max_blocks), "s"(p.num_heads), "s"(p.num_kv_heads), "s"(p.lay.tpb), "s"(p.lay.tpb_log2),
               "s"(p.lay.num_kv_heads), "s"(p.lay.bytes_per_seq), "s"(p.nsplit), "s"(p.split_tokens), "s"(p.rope), "s"(p.rope_max_pos),
               "s"(p.part_ml), "s"(p.part_o), "s"(p.qs.slab), "s"(p.tickets), "s"(p.amax));
  if constexpr (LASTM) {
    if ((int)blockIdx.z >= p.batch) {       // rider workgroups: L2 prefetch of the next projection's weights
      prefetch_weights_to_l2(p.pf, smem, (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)));
      return;
    }
  }
  half_t* q_lds = reinterpret_cast<half_t*>(smem);            // [G][128] dequant order
  half_t* kcur = q_lds + G * DH;                              // [128] natural order (post RoPE)
  half_t* kcur_p = kcur + DH;                                 // [128] dequant order
  half_t* vcur = kcur_p + DH;                                 // [128]
  float* red = reinterpret_cast<float*>(vcur + DH);           // [64]
  int64_t* pages = reinterpret_cast<int64_t*>(red + 64);      // [2][40]
  // G = 8 (head groups of 8 q heads per kv head: Llama-2/3-70B; the MFMAs' 16 head columns hold them at no extra cost, and K / V
  // are fetched and unpacked once for all eight instead of once per four): the waves' O accumulators for the combine would be
  // 16 KiB -- each wave parks them in its OWN V tile instead (it writes them when its sweep is over; nobody else touches that
  // tile), so the workgroup stays at 40 KiB of LDS.  G <= 4 keeps the separate [4 waves][G][128] array.
  constexpr bool XOVER = G > 4;
  float* xbuf = reinterpret_cast<float*>(pages + 80);         // [4 waves][G][128] (G <= 4)
  float* mlbuf = XOVER ? xbuf : xbuf + DEC_WAVES * G * DH;    // [4 waves][G][2]  (max, sum) of each wave
  uint8_t* vtile = reinterpret_cast<uint8_t*>(mlbuf + DEC_WAVES * G * 2);   // [4 waves][32][FVROW]
  static_assert(G <= 16 && G * DH * 4 <= FVTILE, "head columns of the MFMA; accumulators fit the wave's V tile");
  auto xrow = [&](int w, int g) -> float* {
    return XOVER ? reinterpret_cast<float*>(vtile + w * FVTILE) + g * DH : xbuf + ((size_t)w * G + g) * DH;
  };

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x;
  const int group = p.num_heads / p.num_kv_heads;
  const int qg = group / G;
  const int hk = blockIdx.y / qg;
  const int sub = blockIdx.y % qg;
  const int hq0 = hk * group + sub * G;
  const int b = blockIdx.z;
  OMNI_CLK(16);
  const KvLayout lay = p.lay;
  constexpr int RB = KV8 ? DH : ROW_BYTES;   // bytes of one token row of one head
  constexpr int NQ = KV8 ? 2 : 1;            // 16-B pieces per lane and 32 values
  float k_qo = 1.0f, v_qo = 1.0f;
  if constexpr (KV8) { k_qo = p.kv_qo[0]; v_qo = p.kv_qo[1]; }
  // The kernel is a chain of memory round trips, so the requests are ordered to need only two of them:
  //   trip 1 (independent of the sequence length): length, page-table window, raw q / k / v rows;
  //   trip 2: RoPE coefficients of position tlen, and every K and V byte of the split.
  // The split geometry is fixed on the host (split s = virtual tokens [s*split_tokens, +split_tokens)),
  // so the page window does not wait for the length.
  // head class: pool geometry and page table of this kv head
  int hrank = hk, hpool = lay.num_kv_heads, tab_blocks = p.max_blocks;
  bool streaming = false;
  const int64_t* ktab = p.kv_pointers + (size_t)b * 2 * p.max_blocks;
  const int* dyn = nullptr;
  if constexpr (FG) {
    hrank = p.fg.rank[hk];
    streaming = p.fg.flags[hk] == 0;
    hpool = streaming ? p.fg.num_strm : p.fg.num_retr;
    if (streaming) {
      tab_blocks = p.fg.strm_blocks;
      ktab = p.fg.strm_pointers + (size_t)b * 2 * tab_blocks;
    } else if (p.fg.dyn) {
      dyn = p.fg.dyn + ((size_t)b * p.num_heads + hq0) * p.fg.num_dyn;
    }
  }
  const int64_t* vtab = ktab + tab_blocks;
  const int pool_bytes_per_seq = hpool * lay.tpb * RB;
  const float inv_sqrt_dh = 0.08838834764831845f;
  const float sm_scale2 = 0.12751743075284304f;   // log2(e) / sqrt(Dh): the tile sweep keeps its running maxima in log2 units
  const int vt0 = split * p.split_tokens;
  const int page0 = (FG && streaming) ? 0 : (vt0 >> lay.tpb_log2);
  const bool owns_cur = split == p.nsplit - 1;

  // ---- trip 1 -----------------------------------------------------------------------------------------
  int64_t my_page = 0;
  if (tid < 80) {
    const int pi = tid < 40 ? tid : tid - 40;
    const int64_t* tab = tid < 40 ? ktab : vtab;
    const int pg = page0 + pi;
    if constexpr (FG) {
      if (streaming) {               // the whole ring (<= 40 pages, checked on the host)
        if (pg < tab_blocks) my_page = tab[pg];
      } else if (dyn) {              // selected pages, in selection order
        if (pg < p.fg.num_dyn) {
          const int sel = dyn[pg];
          if (sel >= 0 && sel < tab_blocks) my_page = tab[sel];
        }
      } else if (pg < tab_blocks) {
        my_page = tab[pg];
      }
    } else {
#if (OMNI_FLASH_ABLATE & 64)
      my_page = (int64_t)(omni_dbg_pool[tid < 40 ? 0 : 1] + (unsigned long long)(b * p.max_blocks + (pg < p.max_blocks ? pg : 0)) * omni_dbg_pool[2]);
#else
      if (pg < p.max_blocks) my_page = tab[pg];   // entries past the sequence's pages are never dereferenced
#endif
    }
  }
  // the sequence's first K / V page: always allocated (an empty split reads it, unused)
#if (OMNI_FLASH_ABLATE & 64)
  const int64_t dummy_ptr = (int64_t)omni_dbg_pool[tid < 40 ? 0 : 1];
#else
  const int64_t dummy_ptr = tid < 80 ? (tid < 40 ? ktab : vtab)[0] : 0;
#endif
  constexpr int QIT = ((G + 1) * 64 + DEC_THREADS - 1) / DEC_THREADS;
  half_t qa[QIT], qbv[QIT];
  half_t vcur_r = (half_t)0.0f;
  if (p.qs.slab == nullptr) {
#pragma unroll
    for (int j = 0; j < QIT; ++j) {
      const int idx = tid + j * DEC_THREADS;
      const int h = idx >> 6, i = idx & 63;
      const half_t* src = h < G ? p.q + (size_t)b * p.q_stride + (size_t)(hq0 + h) * DH
                                : p.k + (size_t)b * p.kv_stride + (size_t)hk * DH;   // h > G (idle slots): k again
      qa[j] = src[i];
      qbv[j] = src[i + 64];
    }
    if (tid < DH) vcur_r = p.v[(size_t)b * p.kv_stride + (size_t)hk * DH + tid];
  } else {
    // the qkv projection left int32 split-K slabs: sum them and apply its epilogue here (same arithmetic and operation
    // order as qgemm_kernel.h epilogue<>: the values are the fp16 the slab epilogue launch would have stored)
    const QkvSlabSrc qs = p.qs;
    // all loads of the thread's 2 QIT + 1 channels first (up to four slabs each in one batch: a plain loop over the slabs
    // made every slab a dependent L2 round trip in front of everything else the kernel does), then the arithmetic
    constexpr int NE = 2 * QIT + 1, SB = 4;
    int nch[NE];
#pragma unroll
    for (int j = 0; j < QIT; ++j) {
      const int idx = tid + j * DEC_THREADS;
      const int h = idx >> 6, i = idx & 63;
      const int n0 = h < G ? qs.col_q + (hq0 + h) * DH : qs.col_k + hk * DH;
      nch[2 * j] = n0 + i;
      nch[2 * j + 1] = n0 + i + 64;
    }
    nch[NE - 1] = qs.col_v + hk * DH + (tid < DH ? tid : 0);
    int part[NE][SB];
    half_t swh[NE], szh[NE];
    const half_t sah = qs.ascales[b];
    const half_t ash = qs.asum ? qs.asum[b] : (half_t)0.0f;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int32_t* src = qs.slab + (size_t)b * qs.n + nch[e];
#pragma unroll
      for (int k = 0; k < SB; ++k) part[e][k] = src[(size_t)(k < qs.sk ? k : 0) * qs.sstride];
      swh[e] = qs.wscales[nch[e]];
      szh[e] = qs.wsz ? qs.wsz[nch[e]] : (half_t)0.0f;
    }
    const float sa = (float)sah, as = (float)ash;
    half_t val[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      int acc = 0;
#pragma unroll
      for (int k = 0; k < SB; ++k) acc += k < qs.sk ? part[e][k] : 0;
      for (int k = SB; k < qs.sk; ++k) acc += qs.slab[(size_t)k * qs.sstride + (size_t)b * qs.n + nch[e]];
      const float sw = (float)swh[e];
      if (qs.wsz) {
        float t = (float)acc * sw;
        t = t * sa;
        const float c = (float)szh[e] * as;
        val[e] = (half_t)(t - c);
      } else {
        const float sc = sw * sa;
        val[e] = (half_t)((float)acc * sc);
      }
    }
#pragma unroll
    for (int j = 0; j < QIT; ++j) { qa[j] = val[2 * j]; qbv[j] = val[2 * j + 1]; }
    if (tid < DH) vcur_r = val[NE - 1];
  }

  // (sparse) the last selected page is the newest one: requested with trip 1, it bounds the attended tokens below
  int dyn_last = 0;
  if constexpr (FG) {
    if (dyn) dyn_last = dyn[p.fg.num_dyn - 1];
  }
#if (OMNI_FLASH_ABLATE & 64)
  const int tlen = (int)omni_dbg_pool[3];
#else
  const int tlen = p.lengths[b] - 1;
#endif
  int nvirt = tlen, gap = 0;   // attended cached tokens; streaming: virtual i >= sink is token i + gap
  if constexpr (FG) {
    if (streaming) {
      nvirt = min(p.fg.sink + p.fg.local - 1, tlen);
      gap = tlen - nvirt;
    } else if (dyn) {
      // Upstream counts (tlen-1) % tpb + 1 tokens for the last selected page (sparse_attention/...Template.hpp:1568)
      // while its Python layer passes the page of the CURRENT token there (decoding_attention.py:132-142:
      // total_page_num - 1 = timestep // tokens_per_block): when tlen % tpb == 0 that page holds no cached token
      // yet, and upstream then exponentiates score slots it never wrote (:1737-1744 vs :1962).  Here the last
      // page contributes the cached tokens it really holds, clamp(tlen - page * tpb, 0, tpb): identical to
      // upstream's count whenever that is defined, 0 on the boundary step (nothing unwritten is read, no race
      // with the append of the current token into slot 0).
      int in_last = tlen - (dyn_last << lay.tpb_log2);
      in_last = in_last < 0 ? 0 : (in_last > lay.tpb ? lay.tpb : in_last);
      nvirt = (p.fg.num_dyn - 1) * lay.tpb + in_last;
    }
  }
  const int t0 = min(nvirt, vt0);
  const int t1 = min(nvirt, vt0 + p.split_tokens);
  const int nt = t1 - t0;
  // an empty split (nt == 0) still runs one branch-free load batch: it reads slot 0 of window entry 0, which then
  // holds the sequence's first page (always allocated)
  if (tid < 80) pages[tid] = (nt == 0 && (tid == 0 || tid == 40)) ? dummy_ptr : my_page;

  // virtual token -> (index into pages[], slot in the page)
  auto locate = [&](int vt, int& pidx, int& slot) {
    if (FG && streaming) {
      const int lt = vt < p.fg.sink ? vt : vt + gap;
      pidx = ring_block(lt >> lay.tpb_log2, p.fg.sink_blocks, p.fg.local_blocks);
      slot = lt & (lay.tpb - 1);
    } else {
      pidx = (vt >> lay.tpb_log2) - page0;
      slot = vt & (lay.tpb - 1);
    }
  };

  const int ntiles = (nt + 31) >> 5;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int jh = l15 < G ? l15 : 0;                               // this lane's q-head column (clamped)
  const int tail_off = pool_bytes_per_seq + hrank * lay.tpb * 2;
  const int zero_off = hpool * lay.tpb * 2;

  // ---- trip 2: RoPE coefficients + the first K and V batches (normally: all of the split) -------------------
  float rc[QIT], rs[QIT];
  {
    const int rp = tlen < p.rope_max_pos ? tlen : p.rope_max_pos - 1;
    const float* cs = p.rope + (size_t)rp * DH;
#pragma unroll
    for (int j = 0; j < QIT; ++j) {
      const int i = (tid + j * DEC_THREADS) & 63;
      const float2 t = *reinterpret_cast<const float2*>(cs + 2 * i);
      rc[j] = t.x; rs[j] = t.y;
    }
  }
  OMNI_CLK(17);
  __syncthreads();   // pages[] visible
  FLASH_STAMP(1);

  const int vtok = lane >> 2, vpiece = lane & 3;          // coalesced V loads: 4 lanes per token
  const size_t vhead_off = (size_t)hrank * lay.tpb * RB + vpiece * 16 * NQ;
  const size_t khead_off = (size_t)hrank * lay.tpb * RB + l4 * 16 * NQ;  // K: lane = (token l15, 32-value piece l4)
  // One sweep over the split (flash-decoding inside the workgroup): wave w owns the 32-token tiles w, w+4, ... and
  // for each runs Q.K^T -> online softmax in registers -> P.V; a batch of FB tiles' K AND V bytes is in flight
  // while the previous batch is consumed.  No workgroup barrier until the four waves' (max, sum, O) are combined.
  // (fine-grained instantiations keep two tiles per batch: at batch 1 / 256 K tokens the launch has few workgroups and
  //  lives on loads in flight per wave -- 93 vs 98 us for the dense 4 + 4 head mix; sparse decode times the same)
  constexpr int FB = FG ? 2 : OMNI_FLASH_FB;
  // two register sets: the batch being consumed and the batch in flight (PIPE / fine-grained: static indices, the sweep is
  // unrolled by two batches; the short-sweep form copies the arrived batch from set 1 to set 0 and has one loop body)
  uint4 kraw[2][FB][2][NQ], vraw[2][FB][2][NQ];
  half_t ksc[2][FB][2], kze[2][FB][2], vsc[2][FB][2], vze[2][FB][2];   // KV4 only
  // safe token of an out-of-range lane: the split's first token, or (empty split) slot 0 of window entry 0
  const int tsafe = nt > 0 ? t0 : ((FG && streaming) ? 0 : (page0 << lay.tpb_log2));
  // Dense instantiations, long sweeps (PIPE: the host picks it where a wave has six or more tiles): a 16-token group never
  // straddles a page (splits start at multiples of 16 tokens, pages hold a multiple of 16), so its page pointer, first slot and
  // every base address are WAVE-UNIFORM: the page window sits in two VGPR pairs (lane i = window entry i, read with v_readlane),
  // the bases are scalar arithmetic, and the loads take (scalar base, 32-bit lane offset) -- 41 VALU instructions per tile less.
  // Lanes past the split's end re-read its last token (finite values; their scores are masked), a group wholly past it reads
  // the safe token.  Requests run TWO tiles ahead without a third register set: a tile's K registers are dead once its K
  // operands are unpacked and its V registers once the V tile is in LDS, so the K / V requests of tile i + 2 go out right there,
  // into the set tile i is leaving (issued unconditionally: see consume()).  What these bought and what they did not:
  // profiles/r05_b (256 x 2 K tokens 171.7 -> 160 us, 8 x 32 K 88 -> 79; 256 x 1 K unchanged -- the sweep is bound by the
  // unpack arithmetic the reference's per-element fp16 rounding dictates).
  int64_t kwin = 0, vwin = 0;
  if constexpr (!FG && PIPE) {
    kwin = pages[lane < 40 ? lane : 0];
    vwin = pages[40 + (lane < 40 ? lane : 0)];
  }
  auto window_page = [&](int64_t win, int pidx) -> const uint8_t* {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)win, pidx);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)((uint64_t)win >> 32), pidx);
    return reinterpret_cast<const uint8_t*>(((uint64_t)hi << 32) | lo);
  };
  // dense: the K half / V half of one tile's requests (tile u of the batch that starts at tile index i0 of this wave)
  auto load_dense = [&](auto set_tag, auto which_tag, int u, int i0) {
    constexpr int S = decltype(set_tag)::value;
    constexpr int WHICH = decltype(which_tag)::value;      // 0: K bytes + K scales / zeros, 1: V
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int g0 = (wave + DEC_WAVES * (i0 + u)) * 32 + h * 16;      // first token of the group inside the split
      const bool live = g0 < nt;
      const int vt_base = live ? t0 + g0 : tsafe;
      const int lim = live ? min(15, nt - g0 - 1) : 0;                  // last lane token that exists
      const int pidx = (vt_base >> lay.tpb_log2) - page0;
      const size_t row0 = ((size_t)hrank * lay.tpb + (size_t)(vt_base & (lay.tpb - 1))) * RB;
      const size_t tail0 = (size_t)tail_off + 2 * (size_t)(vt_base & (lay.tpb - 1));
      const uint8_t* page = window_page(WHICH == 0 ? kwin : vwin, pidx);
      const uint32_t tok = min((uint32_t)(WHICH == 0 ? l15 : vtok), (uint32_t)lim);
      const uint32_t off = __umul24(tok, (uint32_t)RB) + (uint32_t)((WHICH == 0 ? l4 : vpiece) * 16 * NQ);
#pragma unroll
      for (int n = 0; n < NQ; ++n) {
        if constexpr (WHICH == 0) kraw[S][u][h][n] = gload<uint4>(page + row0 + off + 16 * n);
        else vraw[S][u][h][n] = gload<uint4>(page + row0 + off + 16 * n);
      }
      if constexpr (!KV8) {
        const half_t sc = gload<half_t>(page + tail0 + 2 * tok), ze = gload<half_t>(page + tail0 + zero_off + 2 * tok);
        if constexpr (WHICH == 0) { ksc[S][u][h] = sc; kze[S][u][h] = ze; }
        else { vsc[S][u][h] = sc; vze[S][u][h] = ze; }
      }
    }
  };
  auto load_batch = [&](auto set_tag, int i0) {   // branch-free; page pointers of the whole batch first, then every load
    constexpr int S = decltype(set_tag)::value;
    if constexpr (!FG && PIPE) {      // (short sweeps keep the per-lane addressing: its page lookups do not queue behind one another
#pragma unroll                        //  on the scalar unit in front of a wave's only two batches -- 1 % of the bs = 16 step)
      for (int u = 0; u < FB; ++u) {
        load_dense(set_tag, IntTag<0>{}, u, i0);
        load_dense(set_tag, IntTag<1>{}, u, i0);
      }
      return;
    }
    const uint8_t* kp[FB][2];
    const uint8_t* vp[FB][2];
    int ks_[FB][2], vs_[FB][2];
#pragma unroll
    for (int u = 0; u < FB; ++u)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int tile = wave + DEC_WAVES * (i0 + u);
        int pidx;
        const int tik = tile * 32 + h * 16 + l15;
        locate(tik < nt ? t0 + tik : tsafe, pidx, ks_[u][h]);
        kp[u][h] = reinterpret_cast<const uint8_t*>(pages[pidx]);
        const int tiv = tile * 32 + h * 16 + vtok;
        locate(tiv < nt ? t0 + tiv : tsafe, pidx, vs_[u][h]);
        vp[u][h] = reinterpret_cast<const uint8_t*>(pages[40 + pidx]);
      }
#pragma unroll
    for (int u = 0; u < FB; ++u)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int n = 0; n < NQ; ++n)
          kraw[S][u][h][n] = gload<uint4>(kp[u][h] + khead_off + (size_t)ks_[u][h] * RB + 16 * n);
#pragma unroll
        for (int n = 0; n < NQ; ++n)
          vraw[S][u][h][n] = gload<uint4>(vp[u][h] + vhead_off + (size_t)vs_[u][h] * RB + 16 * n);
        if constexpr (!KV8) {
          const uint8_t* kt = kp[u][h] + tail_off + 2 * ks_[u][h];
          ksc[S][u][h] = gload<half_t>(kt);
          kze[S][u][h] = gload<half_t>(kt + zero_off);
          const uint8_t* vt_ = vp[u][h] + tail_off + 2 * vs_[u][h];
          vsc[S][u][h] = gload<half_t>(vt_);
          vze[S][u][h] = gload<half_t>(vt_ + zero_off);
        }
      }
  };
  load_batch(IntTag<(FG || PIPE) ? 0 : 1>{}, 0);      // (short-sweep form: set 1 is the one in flight, set 0 the one consumed)

  // RoPE(q) (and k of the current token) into LDS while the cache bytes are in flight
#pragma unroll
  for (int j = 0; j < QIT; ++j) {
    const int idx = tid + j * DEC_THREADS;
    const int h = idx >> 6, i = idx & 63;
    if (h > G || (h == G && !owns_cur)) continue;
    const float c = rc[j], sn = rs[j];
    const float a = (float)qa[j], bb = (float)qbv[j];
    const float t0f = c * a, t1f = sn * bb, t2f = c * bb, t3f = sn * a;
    const half_t r0 = (half_t)(t0f - t1f), r1 = (half_t)(t2f + t3f);
    const int p0 = KV8 ? i : perm_pos(i), p1 = KV8 ? i + 64 : perm_pos(i + 64);
    if (h < G) {
      q_lds[h * DH + p0] = r0;
      q_lds[h * DH + p1] = r1;
    } else {
      kcur[i] = r0; kcur[i + 64] = r1;
      kcur_p[p0] = r0; kcur_p[p1] = r1;
    }
  }
  if (owns_cur && tid < DH) vcur[tid] = vcur_r;
  __syncthreads();

  OMNI_CLK(18);
  FLASH_STAMP(2);
  float scur[G];
#pragma unroll
  for (int g = 0; g < G; ++g) scur[g] = 0.0f;
  if (owns_cur) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float a = (float)q_lds[g * DH + lane] * (float)kcur_p[lane] +
                (float)q_lds[g * DH + 64 + lane] * (float)kcur_p[64 + lane];
      a = wave_sum64(a);
      scur[g] = a * inv_sqrt_dh;
    }
  }

  v8h qb[4];  // B operand of S^T = K.q^T: q[head jh][32*l4 + 8s + (0..7 in dequant order)]
#pragma unroll
  for (int sidx = 0; sidx < 4; ++sidx) qb[sidx] = *reinterpret_cast<const v8h*>(q_lds + jh * DH + 32 * l4 + 8 * sidx);
  v4f oacc[8];   // block c: rows = column positions c*16 + 4*l4 + r of the V tile, col = head l15
#pragma unroll
  for (int c = 0; c < 8; ++c) oacc[c] = (v4f){0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.0f;   // this lane's head column: running max, partial sum over its own tokens
  uint8_t* vt = vtile + wave * FVTILE;
  // V tile rows of FVROW = 288 B (72 dwords = 8 mod 64: the 8 rows a 32-lane group of ds_read_b64_tr_b16 touches
  // fall in 8 disjoint 8-bank groups); the 16-B units of pieces 2, 3 are pair-swapped so that the 8 lanes (2 tokens x 4
  // pieces) of a ds_write_b128 group hit 8 different 4-bank groups.  PMC before: 40 % of the LDS cycles were conflicts.
  const int tr_off = (4 * l4 + (l15 >> 2)) * FVROW + (l15 & 3) * 8;
  const int my_tiles = ntiles > wave ? (ntiles - wave + DEC_WAVES - 1) / DEC_WAVES : 0;
  auto consume = [&](auto set_tag, int i0) {
    constexpr int S = decltype(set_tag)::value;
#pragma unroll
    for (int u = 0; u < FB; ++u) {
      const int tbase = (wave + DEC_WAVES * (i0 + u)) * 32;
      if constexpr (!FG && !PIPE) {
        if (tbase >= nt) continue;   // (short-sweep form: no request rides on a tile that does not exist)
      }
      const bool on = tbase < nt;   // wave-uniform: the tile exists.  Its arithmetic is conditional, the REQUESTS below are not:
                                    // vmcnt counts in order, and the compiler sizes every wait for the path with the fewest
                                    // requests behind the one it needs -- a conditional prefetch turns each counted wait into
                                    // (nearly) vmcnt(0), i.e. a wait for the request just issued (ISA of round 5: vmcnt(5 .. 0)
                                    // in front of the V unpack where 17 .. 12 are in flight).  A tile past the end re-reads the
                                    // safe token: two wasted batches per wave, which is why short sweeps do not run this form.
#if (OMNI_FLASH_ABLATE & 1)  // timing experiment (wrong results): the sweep's loads without its arithmetic -- what the launch
      {                      // geometry (pages, workgroups, one batch in flight per wave) can stream at all
        uint32_t acc_x = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          acc_x ^= kraw[S][u][h][0].x ^ kraw[S][u][h][0].y ^ kraw[S][u][h][0].z ^ kraw[S][u][h][0].w;
          acc_x ^= vraw[S][u][h][0].x ^ vraw[S][u][h][0].y ^ vraw[S][u][h][0].z ^ vraw[S][u][h][0].w;
          if constexpr (!KV8) acc_x ^= (uint32_t)__builtin_bit_cast(uint16_t, ksc[S][u][h]) ^ (uint32_t)__builtin_bit_cast(uint16_t, kze[S][u][h]) ^
                                       (uint32_t)__builtin_bit_cast(uint16_t, vsc[S][u][h]) ^ (uint32_t)__builtin_bit_cast(uint16_t, vze[S][u][h]);
        }
        if (on) oacc[0][0] += (float)(acc_x & 1u);
        if constexpr (!FG && PIPE) { load_dense(set_tag, IntTag<0>{}, u, i0 + 2 * FB); load_dense(set_tag, IntTag<1>{}, u, i0 + 2 * FB); }
        continue;
      }
#endif
      // ---- scores of 32 tokens: two 16-token groups ----
      float x[8];
      float tmax = -1e30f;
      if (on) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v2h kd[16];
        if constexpr (KV8) {
          kv8_dequant16(kraw[S][u][h][0], k_qo, kd);
          kv8_dequant16(kraw[S][u][h][NQ - 1], k_qo, kd + 8);
        } else {
          const half_t ch = (half_t)(-(float)ksc[S][u][h] * (float)kze[S][u][h]);
          kv4_dequant16<(OMNI_FLASH_ABLATE & 8) != 0>(kraw[S][u][h][0], (v2h){ksc[S][u][h], ksc[S][u][h]}, (v2h){ch, ch}, kd);
        }
        v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
          const v8h a = {kd[4 * sidx][0], kd[4 * sidx][1], kd[4 * sidx + 1][0], kd[4 * sidx + 1][1],
                         kd[4 * sidx + 2][0], kd[4 * sidx + 2][1], kd[4 * sidx + 3][0], kd[4 * sidx + 3][1]};
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qb[sidx], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)     // acc[r] = q[head l15] . K[token tbase + 16h + 4*l4 + r]; scores in the exp2 domain
          x[4 * h + r] = acc[r] * sm_scale2;
      }
      }
      if constexpr (!FG && PIPE) load_dense(set_tag, IntTag<0>{}, u, i0 + 2 * FB);      // this tile's K registers are free: tile + 2's K
      v8h pb;   // k-slots of the P.V step: tokens 4*l4+r of group 0, then of group 1 (as the V^T operand below)
      if (on) {
      if (tbase + 32 > nt) {   // wave-uniform: only the last tile of a split has token slots to mask (118 -> ~50 softmax VALU per
#pragma unroll                 //  full tile: the sweep is VALU-issue bound at every batch size, profiles/r03_h)
        for (int e = 0; e < 8; ++e) {
          const int ti = tbase + 16 * (e >> 2) + 4 * l4 + (e & 3);
          x[e] = ti < nt ? x[e] : -1e30f;
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) tmax = __builtin_fmaxf(tmax, x[e]);
      tmax = rows4_max(tmax);
      const float m_new = __builtin_fmaxf(m_run, tmax);      // (a tile holds >= 1 real token: m_new is a real score)
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float psum = 0.0f;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        // masked slots: exp2(-1e30 - m_new) = 0.  The row sum takes the unrounded exponentials (as upstream sums them);
        // the P.V operand is their fp16 rounding
        const v2f pp = (OMNI_FLASH_ABLATE & 16) ? (v2f){x[e] - m_new, x[e + 1] - m_new}
                                              : (v2f){__builtin_amdgcn_exp2f(x[e] - m_new), __builtin_amdgcn_exp2f(x[e + 1] - m_new)};
        psum += pp[0] + pp[1];
        const v2h ph = __builtin_convertvector(pp, v2h);
        pb[e] = ph[0];
        pb[e + 1] = ph[1];
      }
      l_run = l_run * alpha + psum;
      if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) oacc[c] *= alpha;
      }
      m_run = m_new;
      // ---- O^T += V^T . P^T : dequantised V tile through LDS, transposed reads ----
      if constexpr ((OMNI_FLASH_ABLATE & 2) != 0) {      // (timing experiment: the V bytes are consumed, nothing else)
        oacc[0][0] += (float)((vraw[S][u][0][0].x ^ vraw[S][u][1][0].y ^ (uint32_t)__builtin_bit_cast(uint16_t, vsc[S][u][0]) ^
                               (uint32_t)__builtin_bit_cast(uint16_t, vze[S][u][1]) ^ (uint32_t)pb[0]) & 1u);
      } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v2h vd[16];
        if constexpr (KV8) {
          kv8_dequant16(vraw[S][u][h][0], v_qo, vd);
          kv8_dequant16(vraw[S][u][h][NQ - 1], v_qo, vd + 8);
        } else {
          const half_t ch = (half_t)(-(float)vsc[S][u][h] * (float)vze[S][u][h]);
          kv4_dequant16<(OMNI_FLASH_ABLATE & 4) != 0>(vraw[S][u][h][0], (v2h){vsc[S][u][h], vsc[S][u][h]}, (v2h){ch, ch}, vd);
        }
        uint8_t* dst = vt + (h * 16 + vtok) * FVROW + vpiece * 64;   // 32 values in dequant order
        const int wsw = vpiece >> 1;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const v8h t = {vd[4 * w][0], vd[4 * w][1], vd[4 * w + 1][0], vd[4 * w + 1][1],
                         vd[4 * w + 2][0], vd[4 * w + 2][1], vd[4 * w + 3][0], vd[4 * w + 3][1]};
          *reinterpret_cast<v8h*>(dst + (w ^ wsw) * 16) = t;
        }
      }
      }
      }
      if constexpr (!FG && PIPE) load_dense(set_tag, IntTag<1>{}, u, i0 + 2 * FB);      // ... and its V registers
      if (on && !(OMNI_FLASH_ABLATE & 2)) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint8_t* src = vt + (c < 4 ? tr_off : (tr_off ^ 16)) + c * 32;   // pieces 2, 3: swapped unit pairs
        const v4hp lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (__attribute__((address_space(3))) v4hp*)(__attribute__((address_space(3))) void*)(src));
        const v4hp hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (__attribute__((address_space(3))) v4hp*)(__attribute__((address_space(3))) void*)(src + 16 * FVROW));
        const v8h a = {(half_t)lo[0], (half_t)lo[1], (half_t)lo[2], (half_t)lo[3],
                       (half_t)hi[0], (half_t)hi[1], (half_t)hi[2], (half_t)hi[3]};
        oacc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb, oacc[c], 0, 0, 0);
      }
      }
    }
    if constexpr (FG) load_batch(set_tag, i0 + 2 * FB);      // fine-grained instantiations: the whole next-but-one batch
    if (i0 < 22) FLASH_STAMP(3 + i0);
  };
  if constexpr (FG || PIPE) {
    load_batch(IntTag<1>{}, FB);      // (batch 0 went out with trip 2)
    for (int i0 = 0; i0 < my_tiles; i0 += 2 * FB) {       // consume() requests batch i0 + 2 FB into the set it empties
      consume(IntTag<0>{}, i0);
      consume(IntTag<1>{}, i0 + FB);
    }
  } else {
    // Short sweeps (bs = 16 at 1 K tokens runs four KV splits: two or three tiles per wave) have little to request two ahead,
    // and the unconditional requests would be two tiles of real loads for nothing, waited for behind the sweep where their
    // registers are reused: +1.4 % on the bs = 16 decode step (same-box A/B).  One tile ahead, conditionally, per-lane addresses.
    for (int i0 = 0; i0 < my_tiles; i0 += FB) {      // one loop body: the arrived batch is copied out of the set in flight
#pragma unroll
      for (int u = 0; u < FB; ++u)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int n = 0; n < NQ; ++n) { kraw[0][u][h][n] = kraw[1][u][h][n]; vraw[0][u][h][n] = vraw[1][u][h][n]; }
          if constexpr (!KV8) {
            ksc[0][u][h] = ksc[1][u][h]; kze[0][u][h] = kze[1][u][h];
            vsc[0][u][h] = vsc[1][u][h]; vze[0][u][h] = vze[1][u][h];
          }
        }
      if (i0 + FB < my_tiles) load_batch(IntTag<1>{}, i0 + FB);   // next batch in flight while this one is consumed
      consume(IntTag<0>{}, i0);
    }
  }
  OMNI_CLK(19);
  FLASH_STAMP(28);
  // ---- combine the four waves' (max, sum, O); add the current token; normalise / emit partials ----------------
  l_run = rows4_sum(l_run);
  if (l15 < G) {
#pragma unroll
    for (int c = 0; c < 8; ++c)
      *reinterpret_cast<v4f*>(xrow(wave, l15) + c * 16 + 4 * l4) = oacc[c];
    if (l4 == 0) {
      mlbuf[(wave * G + l15) * 2 + 0] = m_run * 0.6931471805599453f;    // back to natural units for the combine / merge
      mlbuf[(wave * G + l15) * 2 + 1] = l_run;
    }
  }
  __syncthreads();
  for (int oi = tid; oi < G * DH; oi += DEC_THREADS) {
    const int g = oi >> 7, qpos = oi & 127;
    const int d = KV8 ? qpos : unperm_pos(qpos);
    float M = owns_cur ? scur[g] : -1e30f;
#pragma unroll
    for (int w = 0; w < DEC_WAVES; ++w) M = __builtin_fmaxf(M, mlbuf[(w * G + g) * 2]);
    float acc = 0.0f, L = 0.0f;
#pragma unroll
    for (int w = 0; w < DEC_WAVES; ++w) {
      const float wgt = __expf(mlbuf[(w * G + g) * 2] - M);
      acc += wgt * xrow(w, g)[qpos];
      L += wgt * mlbuf[(w * G + g) * 2 + 1];
    }
    if (owns_cur) {
      const float pc = __expf(scur[g] - M);
      acc += pc * (float)vcur[d];
      L += pc;
    }
    if constexpr (DIRECT) {
      p.out[((size_t)b * p.num_heads + hq0 + g) * DH + d] = (half_t)rounded_f32(acc * (1.0f / (L + 1e-6f)));   // (two roundings: see kv4_decode_merge_kernel)
    } else {
      const size_t pi = ((size_t)b * p.num_heads + hq0 + g) * p.nsplit + split;
      if constexpr (LASTM) {      // written through: another CU's workgroup reads them inside this launch
        __hip_atomic_store(p.part_o + pi * DH + d, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (qpos == 0) {
          __hip_atomic_store(p.part_ml + pi * 2 + 0, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(p.part_ml + pi * 2 + 1, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        p.part_o[pi * DH + d] = acc;
        if (qpos == 0) {
          p.part_ml[pi * 2 + 0] = M;
          p.part_ml[pi * 2 + 1] = L;
        }
      }
    }
  }
  if constexpr (LASTM) {
    // ticket of (sequence b, head group blockIdx.y): the last of the nsplit workgroups merges.  Every storing wave drains
    // its write-through (sc1) stores first; the merging wave reads them with agent-scope (sc1) loads behind the ticket.
    FLASH_STAMP(25);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    FLASH_STAMP(26);
    uint32_t* tk = p.tickets + (size_t)b * gridDim.y + blockIdx.y;
    if (tid == 0) {
      const uint32_t t = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = t == (uint32_t)p.nsplit - 1u;
      if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
      red[0] = last ? 1.0f : 0.0f;
    }
    __syncthreads();
    FLASH_STAMP(27);
    if (red[0] != 0.0f && wave == DEC_WAVES - 1) {      // (the last wave: waves 0 / 1 go on to append the current token)
      const int i = hq0 * DH + lane * VT;                // 8 outputs per lane: G * 128 = up to 512 per workgroup
      if (lane * VT < G * DH) {
        SrcAttnMergeT<true> src{p.part_ml, p.part_o, p.nsplit, p.num_heads, b};      // agent-scope loads: see row_kernels.h
        SrcAttnMergeT<true>::Raw raw;
        float x[VT];
        src.fetch(i, raw);
        src.finish(i, raw, x);
        v8h o;
        float mx = 0.0f;
#pragma unroll
        for (int e = 0; e < VT; ++e) {
          o[e] = (half_t)x[e];                           // x[e] is already an fp16 value
          mx = __builtin_fmaxf(mx, __builtin_fabsf(x[e]));
        }
        *reinterpret_cast<v8h*>(p.out + (size_t)b * p.num_heads * DH + i) = o;
        mx = wave_max64(mx);
        if (lane == 0) amax_raise(p.amax, b, hq0 >> 2, mx);
      }
    }
  }

  OMNI_CLK(22);
  FLASH_STAMP(29);
  // ---- append the current token (quantised) to the cache ------------------------------------------
  if (owns_cur && sub == 0 && wave < 2) {
    const half_t* src = wave == 0 ? kcur : vcur;
    const int64_t* tab = wave == 0 ? ktab : vtab;
    const float x0 = (float)src[lane], x1 = (float)src[64 + lane];
    if constexpr (KV8) {   // static scale, no tail write (dense Template.hpp:1377,2162; the tail code there is commented out)
      const float oq = p.kv_oq[wave];
      int blk8 = tlen >> lay.tpb_log2;
      if (FG && streaming) blk8 = ring_block(blk8, p.fg.sink_blocks, p.fg.local_blocks);
      const int wi8 = blk8 - page0;
      const bool in_window8 = nt > 0 && !(FG && dyn != nullptr) && wi8 >= 0 && wi8 < 40;
      uint8_t* pg8 = reinterpret_cast<uint8_t*>(in_window8 ? pages[(wave == 0 ? 0 : 40) + wi8] : tab[blk8]);
      const int slot8 = tlen & (lay.tpb - 1);
      uint8_t* dst8 = pg8 + ((size_t)hrank * lay.tpb + slot8) * RB;
      dst8[lane] = (uint8_t)rni_sat_s8(oq * x0);
      dst8[64 + lane] = (uint8_t)rni_sat_s8(oq * x1);
      if constexpr (FG) {
        if (wave == 0 && !streaming && p.fg.sub_chunk > 0) {
          const int subs = lay.tpb / p.fg.sub_chunk;
          half_t* kmax = reinterpret_cast<half_t*>(pg8 + pool_bytes_per_seq) + 2 * hpool * lay.tpb +
                         ((size_t)(slot8 / p.fg.sub_chunk) * hpool + hrank) * DH;
          half_t* kmin = kmax + (size_t)subs * hpool * DH;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int d = lane + 64 * h;
            const half_t kv = src[d], omx = kmax[d], omn = kmin[d];
            kmax[d] = (half_t)__builtin_fmaxf((float)omx, (float)kv);
            kmin[d] = (half_t)__builtin_fminf((float)omn, (float)kv);
          }
        }
      }
      return;
    }
    const float mx = wave_max64(__builtin_fmaxf(x0, x1));
    const float mn = -wave_max64(-__builtin_fminf(x0, x1));
    const float range = mx - mn;
    const half_t scale_h = (half_t)(range / 15.0f);
    const float nm = -15.0f * mn;
    const half_t zero_h = (half_t)(nm / range);
    const float inv = 1.0f / (float)scale_h, z = (float)zero_h;
    int blk = tlen >> lay.tpb_log2;
    if (FG && streaming) blk = ring_block(blk, p.fg.sink_blocks, p.fg.local_blocks);
    // the page pointer normally sits in the LDS window already (no dependent load at the kernel's end)
    const int wi = blk - page0;
    const bool in_window = nt > 0 && !(FG && dyn != nullptr) && wi >= 0 && wi < 40;   // nt == 0: entry 0 is the dummy
    uint8_t* pg = reinterpret_cast<uint8_t*>(in_window ? pages[(wave == 0 ? 0 : 40) + wi] : tab[blk]);
    const int slot = tlen & (lay.tpb - 1);
    uint8_t* dst = pg + ((size_t)hrank * lay.tpb + slot) * RB;
    const uint32_t c0 = kv4_code(x0, inv, z), c1 = kv4_code(x1, inv, z);
    const uint32_t n0 = __shfl_down(c0, 1, 64), n1 = __shfl_down(c1, 1, 64);
    if ((lane & 1) == 0) {
      dst[lane >> 1] = (uint8_t)(c0 | (n0 << 4));
      dst[32 + (lane >> 1)] = (uint8_t)(c1 | (n1 << 4));
    }
    if (lane == 0) {
      half_t* scp = reinterpret_cast<half_t*>(pg + pool_bytes_per_seq) + hrank * lay.tpb + slot;
      scp[0] = scale_h;
      scp[hpool * lay.tpb] = zero_h;
    }
    if constexpr (FG) {
      // K pages with min/max statistics: fold the new key into its sub-chunk's indicators
      // (sparse_attention/decoderMaskedMultiheadAttentionTemplate.hpp:1414-1428)
      if (wave == 0 && !streaming && p.fg.sub_chunk > 0) {
        const int subs = lay.tpb / p.fg.sub_chunk;
        half_t* kmax = reinterpret_cast<half_t*>(pg + pool_bytes_per_seq) + 2 * hpool * lay.tpb +
                       ((size_t)(slot / p.fg.sub_chunk) * hpool + hrank) * DH;
        half_t* kmin = kmax + (size_t)subs * hpool * DH;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int d = lane + 64 * h;
          const half_t kv = src[d], omx = kmax[d], omn = kmin[d];
          kmax[d] = (half_t)__builtin_fmaxf((float)omx, (float)kv);
          kmin[d] = (half_t)__builtin_fminf((float)omn, (float)kv);
        }
      }
    }
  }
}

// merge the per-split partials: out = sum_s e^{m_s-M} O_s / (sum_s e^{m_s-M} l_s + 1e-6)
__global__ __launch_bounds__(128) void kv4_decode_merge_kernel(half_t* __restrict__ out,
                                                                const float* __restrict__ part_ml,
                                                                const float* __restrict__ part_o, int nsplit) {
  const size_t bh = blockIdx.x;
  const int d = threadIdx.x;
  float M = -1e30f;
  for (int s = 0; s < nsplit; ++s) M = __builtin_fmaxf(M, part_ml[(bh * nsplit + s) * 2]);
  float l = 0.0f, o = 0.0f;
  for (int s = 0; s < nsplit; ++s) {
    const float w = __expf(part_ml[(bh * nsplit + s) * 2] - M);
    l += w * part_ml[(bh * nsplit + s) * 2 + 1];
    o += w * part_o[(bh * nsplit + s) * DH + d];
  }
  // rounded_f32: product rounded to f32, then to fp16 (two roundings, as the reference's float -> half store); left alone
  // the backend folds the multiply into v_fma_mixlo_f16 (ONE rounding) here but not in the fused merge of
  // elementwise.hip (SrcAttnMerge), and the two paths disagree on ~2^-13 of the elements
  out[bh * DH + d] = (half_t)rounded_f32(o * (1.0f / (l + 1e-6f)));
}

struct DecodePlan {
  int nsplit, split_tokens, g;
  size_t lds_bytes;
};

static thread_local int g_override_nsplit = 0;   // tuning hook, see omni_kv4_decode_set_split_override

static DecodePlan plan_decode(int batch, int num_heads, int num_kv_heads, int max_context, int tokens_per_block,
                              bool per_q_head = false, bool allow_g8 = false) {
  DecodePlan pl;
  const int group = num_heads / num_kv_heads;
  pl.g = group >= 4 ? 4 : group;  // group in {1,2,4,8,...}
  if (allow_g8 && group % 8 == 0) pl.g = 8;      // (the dense two-launch forms: one workgroup serves eight q heads)
  if (per_q_head) pl.g = 1;       // every q head walks its own page list
  const int wgs_per_split = batch * num_kv_heads * (group / pl.g);
  // LDS bound of the two-pass kernel's score buffer; the kernels keep a window of 40 page pointers per split
  const int st_cap = 38 * tokens_per_block < 2048 ? 38 * tokens_per_block : 2048;
  const int max_s0 = (max_context + 63) / 64;
  const int max_s = max_s0 < 1 ? 1 : (max_s0 > DEC_MAX_SPLITS ? DEC_MAX_SPLITS : max_s0);
  auto split_tokens = [&](int n) { return ((max_context + n - 1) / n + 15) & ~15; };
  // Two 256-thread workgroups fit a CU (register file): 512 slots.  The workgroups of a launch run in
  // ceil(W*s / 512) rounds of about (tokens per split + a fixed prologue/epilogue worth ~192 tokens) each; pick
  // the split count that minimises that product (a 17-way split of 64 (sequence, head) pairs was 3 rounds of 1936
  // tokens where 24 splits are 3 rounds of 1376).  A single split skips the merge kernel.
  int s_lo = 1;
  while (s_lo < max_s && split_tokens(s_lo) > st_cap) ++s_lo;
  int s_hi = 2048 / wgs_per_split + 1;
  if (s_hi < 2 * s_lo + 8) s_hi = 2 * s_lo + 8;
  if (s_hi > max_s) s_hi = max_s;
  int s = s_lo;
  long long best = -1;
  for (int n = s_lo; n <= s_hi; ++n) {
    const long long rounds = ((long long)wgs_per_split * n + OMNI_FLASH_SLOTS - 1) / OMNI_FLASH_SLOTS;
    const long long cost = rounds * (split_tokens(n) + 192);
    if (best < 0 || cost < best) { best = cost; s = n; }
  }
  if (g_override_nsplit > 0) s = g_override_nsplit;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  int st = split_tokens(s);
  while (st > st_cap && s < DEC_MAX_SPLITS) {
    ++s;
    st = split_tokens(s);
  }
  if (st > st_cap) st = st_cap;   // caller rejects: max_context > 1024 * st_cap
  pl.nsplit = s;
  pl.split_tokens = st;
  pl.lds_bytes = (size_t)(pl.g + 3) * DH * 2 + 64 * 4 + 80 * 8 + (pl.g > 4 ? 0 : (size_t)DEC_WAVES * pl.g * DH * 4) +
                 (size_t)DEC_WAVES * pl.g * 2 * 4 + (size_t)DEC_WAVES * FVTILE;
  return pl;
}

}  // namespace omni

using namespace omni;

extern "C" int omni_compute_padding_offsets(void* out_i32, const void* cu_seqlens_i32, int batch,
                                            int max_len, int total_tokens, void* stream) {
  if (!out_i32 || !cu_seqlens_i32 || batch < 0 || max_len < 0 || total_tokens < 0) return OMNI_EINVAL;
  if (batch == 0) return OMNI_OK;
  hipLaunchKernelGGL(padding_offsets_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream,
                     (int*)out_i32, (const int*)cu_seqlens_i32, max_len);
  return omni_launch_status();
}

extern "C" int omni_kv4_prefill_write(void* qkv_f16, const void* seq_lens_i32,
                                      const void* padding_offsets_i32, const void* kv_pointers_i64,
                                      int tokens, int batch, int max_blocks, int num_heads,
                                      int num_kv_heads, int head_dim, int max_seq_len,
                                      int tokens_per_block, const void* rope_cos_sin_f32,
                                      int rope_max_pos, int max_position_embeddings, void* stream) {
  if (!qkv_f16 || !seq_lens_i32 || !padding_offsets_i32 || !kv_pointers_i64 || !rope_cos_sin_f32)
    return OMNI_EINVAL;
  if (head_dim != DH || tokens < 0 || batch < 1 || num_heads < 1 || num_kv_heads < 1 ||
      num_heads % num_kv_heads != 0 || tokens_per_block < 16 ||
      (tokens_per_block & (tokens_per_block - 1)) != 0 || rope_max_pos < 1 || max_seq_len < 1)
    return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  const long long slots = (long long)tokens * (num_heads + num_kv_heads);
  const unsigned blocks = (unsigned)((slots + 15) / 16);
  hipLaunchKernelGGL(kv4_prefill_write_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     (half_t*)qkv_f16, (const int*)seq_lens_i32, (const int*)padding_offsets_i32,
                     (const int64_t*)kv_pointers_i64, tokens, max_blocks, num_heads, num_kv_heads,
                     max_seq_len, make_layout(tokens_per_block, num_kv_heads),
                     (const float*)rope_cos_sin_f32, rope_max_pos, max_position_embeddings, FgArgs{},
                     (const float*)nullptr);
  return omni_launch_status();
}

static int fill_fg(FgArgs* fg, const void* streaming_kv_pointers_i64, const void* retrieval_head_flags_i32,
                   const void* head_rank_table_i32, int streaming_blocks, int num_kv_heads,
                   int num_retrieval_kv_heads, int num_streaming_kv_heads, int sink_tokens, int local_tokens,
                   int sink_blocks, int local_blocks) {
  if (!retrieval_head_flags_i32 || !head_rank_table_i32) return OMNI_EINVAL;
  if (num_retrieval_kv_heads < 0 || num_streaming_kv_heads < 0 ||
      num_retrieval_kv_heads + num_streaming_kv_heads != num_kv_heads)
    return OMNI_EINVAL;
  if (num_streaming_kv_heads > 0) {
    if (!streaming_kv_pointers_i64 || sink_tokens < 0 || local_tokens < 1 || sink_blocks < 0 || local_blocks < 1 ||
        streaming_blocks < sink_blocks + local_blocks || sink_blocks + local_blocks > 40)
      return OMNI_EINVAL;
  }
  *fg = FgArgs{};
  fg->strm_pointers = (const int64_t*)streaming_kv_pointers_i64;
  fg->flags = (const int*)retrieval_head_flags_i32;
  fg->rank = (const int*)head_rank_table_i32;
  fg->strm_blocks = streaming_blocks;
  fg->num_retr = num_retrieval_kv_heads;
  fg->num_strm = num_streaming_kv_heads;
  fg->sink = sink_tokens; fg->local = local_tokens;
  fg->sink_blocks = sink_blocks; fg->local_blocks = local_blocks > 0 ? local_blocks : 1;
  return OMNI_OK;
}

static int prefill_write_fg_common(
    void* qkv_f16, const void* seq_lens_i32, const void* padding_offsets_i32, const void* retrieval_kv_pointers_i64,
    const void* streaming_kv_pointers_i64, const void* retrieval_head_flags_i32, const void* head_rank_table_i32,
    int tokens, int batch, int retrieval_blocks, int streaming_blocks, int num_heads, int num_kv_heads,
    int num_retrieval_kv_heads, int num_streaming_kv_heads, int head_dim, int max_seq_len, int tokens_per_block,
    int sink_tokens, int local_tokens, int sink_blocks, int local_blocks, const void* rope_cos_sin_f32,
    int rope_max_pos, int max_position_embeddings, void* stream, const float* kv_oq) {
  if (!qkv_f16 || !seq_lens_i32 || !padding_offsets_i32 || !rope_cos_sin_f32) return OMNI_EINVAL;
  if (head_dim != DH || tokens < 0 || batch < 1 || num_heads < 1 || num_kv_heads < 1 ||
      num_heads % num_kv_heads != 0 || tokens_per_block < 16 ||
      (tokens_per_block & (tokens_per_block - 1)) != 0 || rope_max_pos < 1 || max_seq_len < 1)
    return OMNI_EINVAL;
  FgArgs fg;
  const int rc = fill_fg(&fg, streaming_kv_pointers_i64, retrieval_head_flags_i32, head_rank_table_i32,
                         streaming_blocks, num_kv_heads, num_retrieval_kv_heads, num_streaming_kv_heads, sink_tokens,
                         local_tokens, sink_blocks, local_blocks);
  if (rc != OMNI_OK) return rc;
  if (num_retrieval_kv_heads > 0 && !retrieval_kv_pointers_i64) return OMNI_EINVAL;
  if (tokens == 0) return OMNI_OK;
  const long long slots = (long long)tokens * (num_heads + num_kv_heads);
  const unsigned blocks = (unsigned)((slots + 15) / 16);
  if (kv_oq)
    hipLaunchKernelGGL(kv4_prefill_write_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (half_t*)qkv_f16, (const int*)seq_lens_i32, (const int*)padding_offsets_i32,
                       (const int64_t*)retrieval_kv_pointers_i64, tokens, retrieval_blocks, num_heads, num_kv_heads,
                       max_seq_len, make_layout(tokens_per_block, num_kv_heads),
                       (const float*)rope_cos_sin_f32, rope_max_pos, max_position_embeddings, fg, kv_oq);
  else
    hipLaunchKernelGGL(kv4_prefill_write_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (half_t*)qkv_f16, (const int*)seq_lens_i32, (const int*)padding_offsets_i32,
                       (const int64_t*)retrieval_kv_pointers_i64, tokens, retrieval_blocks, num_heads, num_kv_heads,
                       max_seq_len, make_layout(tokens_per_block, num_kv_heads),
                       (const float*)rope_cos_sin_f32, rope_max_pos, max_position_embeddings, fg, kv_oq);
  return omni_launch_status();
}

extern "C" int omni_kv4_prefill_write_fine_grained(
    void* qkv_f16, const void* seq_lens_i32, const void* padding_offsets_i32, const void* retrieval_kv_pointers_i64,
    const void* streaming_kv_pointers_i64, const void* retrieval_head_flags_i32, const void* head_rank_table_i32,
    int tokens, int batch, int retrieval_blocks, int streaming_blocks, int num_heads, int num_kv_heads,
    int num_retrieval_kv_heads, int num_streaming_kv_heads, int head_dim, int max_seq_len, int tokens_per_block,
    int sink_tokens, int local_tokens, int sink_blocks, int local_blocks, const void* rope_cos_sin_f32,
    int rope_max_pos, int max_position_embeddings, void* stream) {
  return prefill_write_fg_common(qkv_f16, seq_lens_i32, padding_offsets_i32, retrieval_kv_pointers_i64,
                                 streaming_kv_pointers_i64, retrieval_head_flags_i32, head_rank_table_i32, tokens,
                                 batch, retrieval_blocks, streaming_blocks, num_heads, num_kv_heads,
                                 num_retrieval_kv_heads, num_streaming_kv_heads, head_dim, max_seq_len,
                                 tokens_per_block, sink_tokens, local_tokens, sink_blocks, local_blocks,
                                 rope_cos_sin_f32, rope_max_pos, max_position_embeddings, stream, nullptr);
}

// Per-tensor KV8 (LServe's published w8a8kv8 configuration): same head classes and page tables, int8 pages with the
// static scales kv_scale_orig_quant_f32 = device fp32 [2] (K, V).  Replaces
// fused_attention_per_tensor_dense.apply_bias_rope_update_kv_cache (per_tensor_common/update_kv_cache.h:16-44).
extern "C" int omni_kv8_prefill_write_per_tensor(
    void* qkv_f16, const void* kv_scale_orig_quant_f32, const void* seq_lens_i32, const void* padding_offsets_i32,
    const void* retrieval_kv_pointers_i64, const void* streaming_kv_pointers_i64,
    const void* retrieval_head_flags_i32, const void* head_rank_table_i32, int tokens, int batch,
    int retrieval_blocks, int streaming_blocks, int num_heads, int num_kv_heads, int num_retrieval_kv_heads,
    int num_streaming_kv_heads, int head_dim, int max_seq_len, int tokens_per_block, int sink_tokens,
    int local_tokens, int sink_blocks, int local_blocks, const void* rope_cos_sin_f32, int rope_max_pos,
    int max_position_embeddings, void* stream) {
  if (!kv_scale_orig_quant_f32) return OMNI_EINVAL;
  return prefill_write_fg_common(qkv_f16, seq_lens_i32, padding_offsets_i32, retrieval_kv_pointers_i64,
                                 streaming_kv_pointers_i64, retrieval_head_flags_i32, head_rank_table_i32, tokens,
                                 batch, retrieval_blocks, streaming_blocks, num_heads, num_kv_heads,
                                 num_retrieval_kv_heads, num_streaming_kv_heads, head_dim, max_seq_len,
                                 tokens_per_block, sink_tokens, local_tokens, sink_blocks, local_blocks,
                                 rope_cos_sin_f32, rope_max_pos, max_position_embeddings, stream,
                                 (const float*)kv_scale_orig_quant_f32);
}

extern "C" void omni_kv4_decode_set_split_override(int nsplit) {
  // nsplit > 0: force the KV split count of the decode attention (tuning sweeps); 0: the planner decides
  omni::g_override_nsplit = nsplit > 0 ? nsplit : 0;
}

extern "C" size_t omni_kv4_decode_workspace_bytes(int batch, int num_heads, int head_dim, int max_context) {
  (void)head_dim;
  if (batch < 1 || num_heads < 1) return 0;
  // upper bound of the planner's split count for any tokens_per_block >= 16 and any number of kv heads (a split
  // holds >= 512 tokens once the context forces splitting; the round-aware search looks at most up to
  // max(2048 / workgroups-per-split + 1, 2 * minimum split count + 8))
  long long s = 2048 / batch + 1;
  const long long s2 = 2 * (((long long)max_context + 511) / 512 + 1) + 8;
  if (s < s2) s = s2;
  const long long max_s = ((long long)max_context + 63) / 64;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > DEC_MAX_SPLITS) s = DEC_MAX_SPLITS;
  return (size_t)batch * num_heads * (size_t)s * (DH + 2) * sizeof(float);
}

// Fused extension: arm the slab source for the NEXT decode-attention launch of this thread (any flavour; one shot).
static thread_local QkvSlabSrc g_armed_qkv_slabs = {};
static QkvSlabSrc take_armed_qkv_slabs() {
  const QkvSlabSrc q = g_armed_qkv_slabs;
  g_armed_qkv_slabs = QkvSlabSrc{};
  return q;
}
extern "C" int omni_decode_arm_qkv_slabs(const void* slab_i32, int sk, int M, int N, int col_q, int col_k, int col_v,
                                         const void* wscales_f16, const void* ascales_f16, const void* w_szs_f16,
                                         const void* a_ssums_f16) {
  g_armed_qkv_slabs = QkvSlabSrc{};
  if (!slab_i32) return OMNI_OK;      // disarm
  if (sk < 1 || M < 1 || N < 1 || col_q < 0 || col_k < 0 || col_v < 0 || !wscales_f16 || !ascales_f16 ||
      ((w_szs_f16 == nullptr) != (a_ssums_f16 == nullptr)))
    return OMNI_EINVAL;
  QkvSlabSrc q{};
  q.slab = (const int32_t*)slab_i32; q.sstride = (long long)M * N; q.sk = sk; q.n = N;
  q.col_q = col_q; q.col_k = col_k; q.col_v = col_v;
  q.wscales = (const half_t*)wscales_f16; q.wsz = (const half_t*)w_szs_f16;
  q.ascales = (const half_t*)ascales_f16; q.asum = (const half_t*)a_ssums_f16;
  g_armed_qkv_slabs = q;
  return OMNI_OK;
}

static int decode_common(void* out_f16, const void* q_f16, const void* k_f16, const void* v_f16, int64_t q_stride,
                         int64_t kv_stride, const void* kv_pointers_i64, const void* lengths_i32, int batch,
                         int max_blocks, int num_heads, int num_kv_heads, int head_dim, int tokens_per_block,
                         int max_context, const void* rope_cos_sin_f32, int rope_max_pos, void* workspace,
                         size_t workspace_bytes, void* stream, bool partials_only, int* nsplit_out) {
  const QkvSlabSrc armed_qs = take_armed_qkv_slabs();     // (first: an early return must not leave it armed)
  if ((!out_f16 && !partials_only) || !q_f16 || !k_f16 || !v_f16 || !kv_pointers_i64 || !lengths_i32 ||
      !rope_cos_sin_f32 || !workspace)
    return OMNI_EINVAL;
  if (head_dim != DH || batch < 1 || num_heads < 1 || num_kv_heads < 1 || num_heads % num_kv_heads != 0 ||
      tokens_per_block < 16 || (tokens_per_block & (tokens_per_block - 1)) != 0 || max_context < 1 ||
      rope_max_pos < 1)
    return OMNI_EINVAL;
  const int group = num_heads / num_kv_heads;
  if (group != 1 && group != 2 && group % 4 != 0) return OMNI_EINVAL;
  const DecodePlan pl = plan_decode(batch, num_heads, num_kv_heads, max_context, tokens_per_block, false, true);
  if ((long long)pl.nsplit * pl.split_tokens < max_context) return OMNI_EINVAL;
  const size_t need = (size_t)batch * num_heads * pl.nsplit * (DH + 2) * sizeof(float);
  if (workspace_bytes < need) return OMNI_ENOMEM;
  DecodeArgs a;
  a.out = (half_t*)out_f16; a.q = (const half_t*)q_f16; a.k = (const half_t*)k_f16; a.v = (const half_t*)v_f16;
  a.q_stride = q_stride; a.kv_stride = kv_stride;
  a.kv_pointers = (const int64_t*)kv_pointers_i64; a.lengths = (const int*)lengths_i32;
  a.batch = batch; a.max_blocks = max_blocks; a.num_heads = num_heads; a.num_kv_heads = num_kv_heads;
  a.lay = make_layout(tokens_per_block, num_kv_heads);
  a.nsplit = pl.nsplit; a.split_tokens = pl.split_tokens;
  a.rope = (const float*)rope_cos_sin_f32; a.rope_max_pos = rope_max_pos;
  a.part_ml = (float*)workspace;
  a.part_o = a.part_ml + (size_t)batch * num_heads * pl.nsplit * 2;
  a.fg = FgArgs{};
  a.kv_qo = nullptr; a.kv_oq = nullptr;
  a.qs = armed_qs;
  if (a.qs.slab && (a.qs.sstride != (long long)batch * a.qs.n || a.qs.col_q + num_heads * DH > a.qs.n ||
                    a.qs.col_k + num_kv_heads * DH > a.qs.n || a.qs.col_v + num_kv_heads * DH > a.qs.n))
    return OMNI_EINVAL;
  dim3 grid(pl.nsplit, num_kv_heads * (group / pl.g), batch);
  hipStream_t st = (hipStream_t)stream;
  if (pl.lds_bytes > 160 * 1024) return OMNI_EINVAL;
  const bool pipe = pl.split_tokens > 5 * 32 * DEC_WAVES;      // a wave sweeps six or more 32-token tiles (the two batches the
                                                               // pipelined form requests past the end are real loads: see the kernel)
#define OMNI_LAUNCH_DEC(G_, D_)                                                                                          \
  do {                                                                                                                   \
    if (pipe) hipLaunchKernelGGL((kv4_decode_flash_kernel<G_, D_>), grid, dim3(DEC_THREADS), pl.lds_bytes, st, a);       \
    else hipLaunchKernelGGL((kv4_decode_flash_kernel<G_, D_, false, false, false, false>), grid, dim3(DEC_THREADS), pl.lds_bytes, st, a); \
  } while (0)
  if (pl.nsplit == 1 && !partials_only) {
    switch (pl.g) {
      case 1: OMNI_LAUNCH_DEC(1, true); break;
      case 2: OMNI_LAUNCH_DEC(2, true); break;
      case 8: OMNI_LAUNCH_DEC(8, true); break;
      default: OMNI_LAUNCH_DEC(4, true); break;
    }
  } else {
    switch (pl.g) {
      case 1: OMNI_LAUNCH_DEC(1, false); break;
      case 2: OMNI_LAUNCH_DEC(2, false); break;
      case 8: OMNI_LAUNCH_DEC(8, false); break;
      default: OMNI_LAUNCH_DEC(4, false); break;
    }
    if (!partials_only)
      hipLaunchKernelGGL(kv4_decode_merge_kernel, dim3(batch * num_heads), dim3(128), 0, st, (half_t*)out_f16,
                         a.part_ml, a.part_o, pl.nsplit);
  }
#undef OMNI_LAUNCH_DEC
  if (nsplit_out) *nsplit_out = pl.nsplit;
  return omni_launch_status();
}

// Fused extension: omni_kv4_decode_attention_partial + omni_attn_merge_f16_amax as ONE launch (the last-arriving split
// workgroup of every (sequence, head group) merges: LASTM above).  `tickets`: batch * num_kv_heads * (group / G) uint32 words,
// zero on entry, zero again on exit.  Falls back to the two launches when the plan has a single split.  *launches_out: 1 or 2.
namespace omni { PrefetchArgs take_armed_prefetch(); }      // qgemm_plan.hip
extern "C" int omni_attn_merge_f16_amax(void* out_f16, const void* part_ml_f32, const void* part_o_f32, int nsplit,
                                        void* amax_slots_u32, int batch, int num_heads, void* stream);
extern "C" int omni_kv4_decode_attention_f16_amax(void* out_f16, void* amax_slots_u32, const void* q_f16, const void* k_f16,
                                                  const void* v_f16, int64_t q_stride, int64_t kv_stride,
                                                  const void* kv_pointers_i64, const void* lengths_i32, int batch,
                                                  int max_blocks, int num_heads, int num_kv_heads, int head_dim,
                                                  int tokens_per_block, int max_context, const void* rope_cos_sin_f32,
                                                  int rope_max_pos, void* workspace, size_t workspace_bytes, void* tickets_u32,
                                                  size_t tickets_words, void* stream) {
  if (!out_f16 || !amax_slots_u32 || !tickets_u32) return OMNI_EINVAL;
  if (num_heads < 1 || num_kv_heads < 1 || num_heads % num_kv_heads != 0 || num_heads % 4 != 0 || batch < 1 || batch > AMAX_ROWS)
    return OMNI_EINVAL;
  const int group = num_heads / num_kv_heads;
  const DecodePlan pl = plan_decode(batch, num_heads, num_kv_heads, max_context < 1 ? 1 : max_context,
                                    tokens_per_block < 16 ? 16 : tokens_per_block);
  if (pl.nsplit < 2 || pl.g != 4 || group % 4 != 0) {      // one split (or an odd head group): the two-launch form
    int ns = 0;
    const int rc = decode_common(nullptr, q_f16, k_f16, v_f16, q_stride, kv_stride, kv_pointers_i64, lengths_i32, batch, max_blocks,
                                 num_heads, num_kv_heads, head_dim, tokens_per_block, max_context, rope_cos_sin_f32,
                                 rope_max_pos, workspace, workspace_bytes, stream, true, &ns);
    if (rc != OMNI_OK) return rc;
    const size_t ml = (size_t)batch * num_heads * ns * 2 * sizeof(float);
    return omni_attn_merge_f16_amax(out_f16, workspace, (const uint8_t*)workspace + ml, ns, amax_slots_u32, batch, num_heads, stream);
  }
  const QkvSlabSrc armed_qs = take_armed_qkv_slabs();
  if (!q_f16 || !k_f16 || !v_f16 || !kv_pointers_i64 || !lengths_i32 || !rope_cos_sin_f32 || !workspace) return OMNI_EINVAL;
  if (head_dim != DH || tokens_per_block < 16 || (tokens_per_block & (tokens_per_block - 1)) != 0 || max_context < 1 ||
      rope_max_pos < 1)
    return OMNI_EINVAL;
  if ((long long)pl.nsplit * pl.split_tokens < max_context) return OMNI_EINVAL;
  const size_t need = (size_t)batch * num_heads * pl.nsplit * (DH + 2) * sizeof(float);
  if (workspace_bytes < need) return OMNI_ENOMEM;
  const int gy = num_kv_heads * (group / pl.g);
  if (tickets_words < (size_t)batch * gy) return OMNI_ENOMEM;
  DecodeArgs a;
  a.out = (half_t*)out_f16; a.q = (const half_t*)q_f16; a.k = (const half_t*)k_f16; a.v = (const half_t*)v_f16;
  a.q_stride = q_stride; a.kv_stride = kv_stride;
  a.kv_pointers = (const int64_t*)kv_pointers_i64; a.lengths = (const int*)lengths_i32;
  a.batch = batch; a.max_blocks = max_blocks; a.num_heads = num_heads; a.num_kv_heads = num_kv_heads;
  a.lay = make_layout(tokens_per_block, num_kv_heads);
  a.nsplit = pl.nsplit; a.split_tokens = pl.split_tokens;
  a.rope = (const float*)rope_cos_sin_f32; a.rope_max_pos = rope_max_pos;
  a.part_ml = (float*)workspace;
  a.part_o = a.part_ml + (size_t)batch * num_heads * pl.nsplit * 2;
  a.fg = FgArgs{};
  a.kv_qo = nullptr; a.kv_oq = nullptr;
  a.qs = armed_qs;
  if (a.qs.slab && (a.qs.sstride != (long long)batch * a.qs.n || a.qs.col_q + num_heads * DH > a.qs.n ||
                    a.qs.col_k + num_kv_heads * DH > a.qs.n || a.qs.col_v + num_kv_heads * DH > a.qs.n))
    return OMNI_EINVAL;
  a.tickets = (uint32_t*)tickets_u32; a.amax = (uint32_t*)amax_slots_u32;
  a.pf = take_armed_prefetch();
  {
    // The fetching workgroups wait ~6 us (12 x s_sleep 16) before they start: the attention workgroups' first two load trips
    // (length / page table / q, then every K / V byte of the split) are latency chains that the weight stream of 160 fetching
    // workgroups slowed down, and the o projection's 8.4 MB need 2.5 of the launch's ~13 us (step -1.2 %, profiles/r04_e).
    // Short splits (a short launch): 2 us.
    static const int attn_pf_delay = omni_knob("OMNI_ATTN_PF_DELAY", -1);
    a.pf.delay = attn_pf_delay >= 0 ? attn_pf_delay : (pl.split_tokens >= 128 ? 12 : 4);
  }
  const int per_slice = pl.nsplit * gy;
  int rider_slices = 0;
  if (a.pf.blocks > 0) {
    rider_slices = (a.pf.blocks + per_slice - 1) / per_slice;
    a.pf.blocks = rider_slices * per_slice;               // whole z slices
    a.pf.first_block = per_slice * batch;
  }
  if (pl.lds_bytes > 160 * 1024) return OMNI_EINVAL;
  dim3 grid(pl.nsplit, gy, batch + rider_slices);
  if (pl.split_tokens > 5 * 32 * DEC_WAVES)
    hipLaunchKernelGGL((kv4_decode_flash_kernel<4, false, false, false, true>), grid, dim3(DEC_THREADS), pl.lds_bytes,
                       (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((kv4_decode_flash_kernel<4, false, false, false, true, false>), grid, dim3(DEC_THREADS), pl.lds_bytes,
                       (hipStream_t)stream, a);
  return omni_launch_status();
}

extern "C" int omni_kv4_decode_attention(void* out_f16, const void* q_f16, const void* k_f16,
                                         const void* v_f16, int64_t q_stride, int64_t kv_stride,
                                         const void* kv_pointers_i64, const void* lengths_i32,
                                         int batch, int max_blocks, int num_heads, int num_kv_heads,
                                         int head_dim, int tokens_per_block, int max_context,
                                         const void* rope_cos_sin_f32, int rope_max_pos,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  return decode_common(out_f16, q_f16, k_f16, v_f16, q_stride, kv_stride, kv_pointers_i64, lengths_i32, batch,
                       max_blocks, num_heads, num_kv_heads, head_dim, tokens_per_block, max_context,
                       rope_cos_sin_f32, rope_max_pos, workspace, workspace_bytes, stream, false, nullptr);
}

// Fused extension: same as omni_kv4_decode_attention but stops after the per-split partials
// (workspace = part_ml f32 [B,Hq,S,2] | part_o f32 [B,Hq,S,128], *nsplit_out = S); the merge is then fused
// with the following per-token quantisation by omni_attn_merge_quant_fuse_sum.
extern "C" int omni_kv4_decode_attention_partial(const void* q_f16, const void* k_f16, const void* v_f16,
                                                 int64_t q_stride, int64_t kv_stride, const void* kv_pointers_i64,
                                                 const void* lengths_i32, int batch, int max_blocks, int num_heads,
                                                 int num_kv_heads, int head_dim, int tokens_per_block,
                                                 int max_context, const void* rope_cos_sin_f32, int rope_max_pos,
                                                 void* workspace, size_t workspace_bytes, int* nsplit_out,
                                                 void* stream) {
  if (!nsplit_out) return OMNI_EINVAL;
  return decode_common(nullptr, q_f16, k_f16, v_f16, q_stride, kv_stride, kv_pointers_i64, lengths_i32, batch,
                       max_blocks, num_heads, num_kv_heads, head_dim, tokens_per_block, max_context,
                       rope_cos_sin_f32, rope_max_pos, workspace, workspace_bytes, stream, true, nsplit_out);
}

// LServe fine-grained decode attention: retrieval heads (optionally restricted to the pages in
// dynamic_sparse_page_idx [B,Hq,num_dynamic_pages]) + streaming heads; replaces
// fused_attention_fine_grained_{dense,sparse}.single_query_attention (KV4 with zeros).
static int decode_fg_common(
    void* out_f16, const void* q_f16, const void* k_f16, const void* v_f16, int64_t q_stride, int64_t kv_stride,
    const void* retrieval_kv_pointers_i64, const void* streaming_kv_pointers_i64,
    const void* retrieval_head_flags_i32, const void* head_rank_table_i32, const void* lengths_i32,
    const void* dynamic_sparse_page_idx_i32, int num_dynamic_pages, int tokens_per_sub_chunk, int batch,
    int retrieval_blocks, int streaming_blocks, int num_heads, int num_kv_heads, int num_retrieval_kv_heads,
    int num_streaming_kv_heads, int head_dim, int tokens_per_block, int sink_tokens, int local_tokens,
    int sink_blocks, int local_blocks, int max_context, const void* rope_cos_sin_f32, int rope_max_pos,
    void* workspace, size_t workspace_bytes, void* stream, const float* kv_qo, const float* kv_oq,
    bool partials_only = false, int* nsplit_out = nullptr) {
  const QkvSlabSrc armed_qs = take_armed_qkv_slabs();     // (first: an early return must not leave it armed)
  if ((!out_f16 && !partials_only) || !q_f16 || !k_f16 || !v_f16 || !lengths_i32 || !rope_cos_sin_f32 || !workspace)
    return OMNI_EINVAL;
  if (head_dim != DH || batch < 1 || num_heads < 1 || num_kv_heads < 1 || num_heads % num_kv_heads != 0 ||
      tokens_per_block < 16 || (tokens_per_block & (tokens_per_block - 1)) != 0 || max_context < 1 ||
      rope_max_pos < 1)
    return OMNI_EINVAL;
  const int group = num_heads / num_kv_heads;
  if (group != 1 && group != 2 && group % 4 != 0) return OMNI_EINVAL;
  DecodeArgs a;
  a.qs = armed_qs;
  int rc = fill_fg(&a.fg, streaming_kv_pointers_i64, retrieval_head_flags_i32, head_rank_table_i32, streaming_blocks,
                   num_kv_heads, num_retrieval_kv_heads, num_streaming_kv_heads, sink_tokens, local_tokens,
                   sink_blocks, local_blocks);
  if (rc != OMNI_OK) return rc;
  if (num_retrieval_kv_heads > 0 && !retrieval_kv_pointers_i64) return OMNI_EINVAL;
  const bool sparse = dynamic_sparse_page_idx_i32 != nullptr;
  if (sparse && (num_dynamic_pages < 1 || tokens_per_sub_chunk < 1 || tokens_per_block % tokens_per_sub_chunk != 0))
    return OMNI_EINVAL;
  a.fg.dyn = (const int*)dynamic_sparse_page_idx_i32;
  a.fg.num_dyn = sparse ? num_dynamic_pages : 0;
  a.fg.sub_chunk = sparse ? tokens_per_sub_chunk : 0;
  // longest attended token list of any head: retrieval heads the whole history (or the selected pages), streaming
  // heads sink + local - 1 tokens -- which can exceed a small page budget (the splits must cover both)
  int span = sparse ? num_dynamic_pages * tokens_per_block : max_context;
  if (num_streaming_kv_heads > 0) {
    const int strm = sink_tokens + local_tokens < max_context ? sink_tokens + local_tokens : max_context;
    if (num_retrieval_kv_heads == 0) span = strm;
    else if (strm > span) span = strm;
  }
  if (span > max_context && !sparse) span = max_context;
  if (span < 1) span = 1;
  const DecodePlan pl = plan_decode(batch, num_heads, num_kv_heads, span, tokens_per_block, sparse);
  if ((long long)pl.nsplit * pl.split_tokens < span) return OMNI_EINVAL;
  if (pl.lds_bytes > 160 * 1024) return OMNI_EINVAL;
  const size_t need = (size_t)batch * num_heads * pl.nsplit * (DH + 2) * sizeof(float);
  if (workspace_bytes < need) return OMNI_ENOMEM;
  a.out = (half_t*)out_f16; a.q = (const half_t*)q_f16; a.k = (const half_t*)k_f16; a.v = (const half_t*)v_f16;
  a.q_stride = q_stride; a.kv_stride = kv_stride;
  a.kv_pointers = (const int64_t*)retrieval_kv_pointers_i64; a.lengths = (const int*)lengths_i32;
  a.batch = batch; a.max_blocks = retrieval_blocks; a.num_heads = num_heads; a.num_kv_heads = num_kv_heads;
  a.lay = make_layout(tokens_per_block, num_kv_heads);
  a.nsplit = pl.nsplit; a.split_tokens = pl.split_tokens;
  a.rope = (const float*)rope_cos_sin_f32; a.rope_max_pos = rope_max_pos;
  a.part_ml = (float*)workspace;
  a.part_o = a.part_ml + (size_t)batch * num_heads * pl.nsplit * 2;
  dim3 grid(pl.nsplit, num_kv_heads * (group / pl.g), batch);
  hipStream_t st = (hipStream_t)stream;
  a.kv_qo = kv_qo; a.kv_oq = kv_oq;
  if (a.qs.slab && (a.qs.sstride != (long long)batch * a.qs.n || a.qs.col_q + num_heads * DH > a.qs.n ||
                    a.qs.col_k + num_kv_heads * DH > a.qs.n || a.qs.col_v + num_kv_heads * DH > a.qs.n))
    return OMNI_EINVAL;
#define OMNI_LAUNCH_FG(G_)                                                                              \
  do {                                                                                                  \
    if (kv_qo)                                                                                          \
      hipLaunchKernelGGL((kv4_decode_flash_kernel<G_, false, true, true>), grid, dim3(DEC_THREADS), pl.lds_bytes, st, a); \
    else                                                                                                \
      hipLaunchKernelGGL((kv4_decode_flash_kernel<G_, false, true>), grid, dim3(DEC_THREADS), pl.lds_bytes, st, a); \
  } while (0)
  switch (pl.g) {
    case 1: OMNI_LAUNCH_FG(1); break;
    case 2: OMNI_LAUNCH_FG(2); break;
    default: OMNI_LAUNCH_FG(4); break;
  }
#undef OMNI_LAUNCH_FG
  if (!partials_only)
    hipLaunchKernelGGL(kv4_decode_merge_kernel, dim3(batch * num_heads), dim3(128), 0, st, (half_t*)out_f16, a.part_ml,
                       a.part_o, pl.nsplit);
  if (nsplit_out) *nsplit_out = pl.nsplit;
  return omni_launch_status();
}

// Fused extension (LServe decode): the fine-grained / per-tensor decode attention without its merge step
// (partials as omni_kv4_decode_attention_partial leaves them); omni_attn_merge_quant_fuse_sum finishes.
// kv_scale_*_f32 == NULL: KV4 pages (fine_grained), else per-tensor KV8 pages.
extern "C" int omni_kv_decode_attention_fine_grained_partial(
    const void* q_f16, const void* k_f16, const void* v_f16, int64_t q_stride, int64_t kv_stride,
    const void* kv_scale_quant_orig_f32, const void* kv_scale_orig_quant_f32,
    const void* retrieval_kv_pointers_i64, const void* streaming_kv_pointers_i64,
    const void* retrieval_head_flags_i32, const void* head_rank_table_i32, const void* lengths_i32,
    const void* dynamic_sparse_page_idx_i32, int num_dynamic_pages, int tokens_per_sub_chunk, int batch,
    int retrieval_blocks, int streaming_blocks, int num_heads, int num_kv_heads, int num_retrieval_kv_heads,
    int num_streaming_kv_heads, int head_dim, int tokens_per_block, int sink_tokens, int local_tokens,
    int sink_blocks, int local_blocks, int max_context, const void* rope_cos_sin_f32, int rope_max_pos,
    void* workspace, size_t workspace_bytes, int* nsplit_out, void* stream) {
  if (!nsplit_out || ((kv_scale_quant_orig_f32 == nullptr) != (kv_scale_orig_quant_f32 == nullptr))) return OMNI_EINVAL;
  return decode_fg_common(nullptr, q_f16, k_f16, v_f16, q_stride, kv_stride, retrieval_kv_pointers_i64,
                          streaming_kv_pointers_i64, retrieval_head_flags_i32, head_rank_table_i32, lengths_i32,
                          dynamic_sparse_page_idx_i32, num_dynamic_pages, tokens_per_sub_chunk, batch,
                          retrieval_blocks, streaming_blocks, num_heads, num_kv_heads, num_retrieval_kv_heads,
                          num_streaming_kv_heads, head_dim, tokens_per_block, sink_tokens, local_tokens, sink_blocks,
                          local_blocks, max_context, rope_cos_sin_f32, rope_max_pos, workspace, workspace_bytes,
                          stream, (const float*)kv_scale_quant_orig_f32, (const float*)kv_scale_orig_quant_f32, true,
                          nsplit_out);
}

OMNI_CLK_READER(omni_debug_clocks_kv)
#ifdef OMNI_DEBUG_CLOCKS
extern "C" int omni_debug_timeline_flash(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(omni::omni_dbg_flash), sizeof(unsigned long long) * 3 * omni::DEC_WAVES * 32) == hipSuccess ? 0 : -5;
}
#endif

extern "C" int omni_kv4_decode_attention_fine_grained(
    void* out_f16, const void* q_f16, const void* k_f16, const void* v_f16, int64_t q_stride, int64_t kv_stride,
    const void* retrieval_kv_pointers_i64, const void* streaming_kv_pointers_i64,
    const void* retrieval_head_flags_i32, const void* head_rank_table_i32, const void* lengths_i32,
    const void* dynamic_sparse_page_idx_i32, int num_dynamic_pages, int tokens_per_sub_chunk, int batch,
    int retrieval_blocks, int streaming_blocks, int num_heads, int num_kv_heads, int num_retrieval_kv_heads,
    int num_streaming_kv_heads, int head_dim, int tokens_per_block, int sink_tokens, int local_tokens,
    int sink_blocks, int local_blocks, int max_context, const void* rope_cos_sin_f32, int rope_max_pos,
    void* workspace, size_t workspace_bytes, void* stream) {
  return decode_fg_common(out_f16, q_f16, k_f16, v_f16, q_stride, kv_stride, retrieval_kv_pointers_i64,
                          streaming_kv_pointers_i64, retrieval_head_flags_i32, head_rank_table_i32, lengths_i32,
                          dynamic_sparse_page_idx_i32, num_dynamic_pages, tokens_per_sub_chunk, batch,
                          retrieval_blocks, streaming_blocks, num_heads, num_kv_heads, num_retrieval_kv_heads,
                          num_streaming_kv_heads, head_dim, tokens_per_block, sink_tokens, local_tokens, sink_blocks,
                          local_blocks, max_context, rope_cos_sin_f32, rope_max_pos, workspace, workspace_bytes,
                          stream, nullptr, nullptr);
}

// Per-tensor KV8 decode attention: replaces fused_attention_per_tensor_{dense,sparse}.single_query_attention
// (fused_attention_per_tensor/dense_attention/fused_attention.h:18-46, sparse_attention/fused_attention.h:18-50).
// kv_scale_quant_orig_f32 / kv_scale_orig_quant_f32: device fp32 [2] (K, V), as the reference's wrappers pass them.
extern "C" int omni_kv8_decode_attention_per_tensor(
    void* out_f16, const void* q_f16, const void* k_f16, const void* v_f16, int64_t q_stride, int64_t kv_stride,
    const void* kv_scale_quant_orig_f32, const void* kv_scale_orig_quant_f32,
    const void* retrieval_kv_pointers_i64, const void* streaming_kv_pointers_i64,
    const void* retrieval_head_flags_i32, const void* head_rank_table_i32, const void* lengths_i32,
    const void* dynamic_sparse_page_idx_i32, int num_dynamic_pages, int tokens_per_sub_chunk, int batch,
    int retrieval_blocks, int streaming_blocks, int num_heads, int num_kv_heads, int num_retrieval_kv_heads,
    int num_streaming_kv_heads, int head_dim, int tokens_per_block, int sink_tokens, int local_tokens,
    int sink_blocks, int local_blocks, int max_context, const void* rope_cos_sin_f32, int rope_max_pos,
    void* workspace, size_t workspace_bytes, void* stream) {
  if (!kv_scale_quant_orig_f32 || !kv_scale_orig_quant_f32) return OMNI_EINVAL;
  return decode_fg_common(out_f16, q_f16, k_f16, v_f16, q_stride, kv_stride, retrieval_kv_pointers_i64,
                          streaming_kv_pointers_i64, retrieval_head_flags_i32, head_rank_table_i32, lengths_i32,
                          dynamic_sparse_page_idx_i32, num_dynamic_pages, tokens_per_sub_chunk, batch,
                          retrieval_blocks, streaming_blocks, num_heads, num_kv_heads, num_retrieval_kv_heads,
                          num_streaming_kv_heads, head_dim, tokens_per_block, sink_tokens, local_tokens, sink_blocks,
                          local_blocks, max_context, rope_cos_sin_f32, rope_max_pos, workspace, workspace_bytes,
                          stream, (const float*)kv_scale_quant_orig_f32, (const float*)kv_scale_orig_quant_f32);
}
