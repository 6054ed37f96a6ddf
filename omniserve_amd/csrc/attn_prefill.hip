// Varlen causal prefill attention (dense heads and token-streaming "Lambda" heads) for MI355X (gfx950).
//
// Replaces the un-vendored third-party kernels the reference calls for ALL prefill attention
// (omniserve/modeling/layers/ctx_attn/ctx_attn_func.py:39-45,68-73):
//   block_sparse_attn.flash_attn_varlen_func(q,k,v,cu_q,cu_k,max_q,max_k,dropout_p=0,causal=True)
//   block_sparse_attn.token_streaming_attn_func(q,k,v,cu_q,cu_k,head_mask_type,streaming_info,max_q,max_k)
// Semantics (inferred, the package is not vendored): softmax(q k^T / sqrt(Dh) + mask) v per
// sequence and q head, GQA by head index division; mask = causal (bottom-right aligned) and, for heads with
// head_mask_type < 0, additionally (k_pos < sink  OR  q_pos - k_pos < local).
//
// Structure (flash-attention, online softmax, matrix cores): a wave owns 16 query rows and walks the
// keys 32 at a time.  It computes S^T = K Q^T (A = K rows straight from global memory, B = Q held in
// registers) so that a lane ends up with 8 keys of ONE query row -- exactly the B-operand layout of the
// second product O^T = V^T P^T, whose A operand (V transposed) comes from an fp16 V tile staged in LDS
// and read back with ds_read_b64_tr_b16.  No shuffles are needed to turn scores into probabilities'
// operand layout; row statistics reduce over the 4 lanes that share a query row.
#include "common.h"

namespace omni {

constexpr int PDH = 128;
constexpr int PVROW = 272;            // bytes per key row of the V tile in LDS (256 + 16 pad)
constexpr int PVTILE = 32 * PVROW;
constexpr int PWAVES = 4;             // 4 waves x 16 query rows = 64 rows per workgroup
typedef __fp16 pv4hp __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct PrefillArgs {
  const half_t* q; const half_t* k; const half_t* v; half_t* out;
  int64_t q_stride, k_stride, v_stride;     // elements between consecutive tokens
  const int* cu_q; const int* cu_k;
  const int* head_mask_type;                 // [Hq] or null (all dense)
  const int* streaming_info;                 // [2*Hq] (sink, local) or null
  int num_heads, num_kv_heads;
  int causal;
};

__global__ __launch_bounds__(64 * PWAVES) void prefill_attn_kernel(PrefillArgs p) {
  __shared__ __attribute__((aligned(16))) uint8_t vtile[PWAVES * PVTILE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int hk = h / (p.num_heads / p.num_kv_heads);
  const int q_begin = p.cu_q[b], len_q = p.cu_q[b + 1] - q_begin;
  const int k_begin = p.cu_k[b], len_k = p.cu_k[b + 1] - k_begin;
  const int q0 = blockIdx.x * (16 * PWAVES) + wave * 16;      // first query row of this wave
  if (blockIdx.x * (16 * PWAVES) >= len_q) return;              // whole workgroup out of range
  if (q0 >= len_q) return;                                      // waves are independent (private LDS tiles, no barriers)
  const int off = len_k - len_q;                                // bottom-right aligned causal mask
  const bool streaming = p.head_mask_type != nullptr && p.head_mask_type[h] < 0;
  const int sink = streaming ? p.streaming_info[2 * h] : 0;
  const int local = streaming ? p.streaming_info[2 * h + 1] : 0;
  const float scale = 0.08838834764831845f;                     // 1/sqrt(128)

  const int qrow = q0 + l15;                                    // this lane's query row (column j of S^T)
  const int qr_c = qrow < len_q ? qrow : (len_q - 1);
  // B operand of S^T = K Q^T: Q[qrow][32s + 8*l4 + (0..7)]
  v8h qb[4];
  {
    const half_t* qp = p.q + (size_t)(q_begin + qr_c) * p.q_stride + (size_t)h * PDH + 8 * l4;
#pragma unroll
    for (int s = 0; s < 4; ++s) qb[s] = *reinterpret_cast<const v8h*>(qp + 32 * s);
  }
  v4f oacc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) oacc[c] = (v4f){0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.0f;

  // key range this workgroup needs
  const int q_last = min(blockIdx.x * (16 * PWAVES) + 16 * PWAVES, len_q) - 1;   // last row of the WG
  const int k_hi = p.causal ? min(len_k, q_last + off + 1) : len_k;             // exclusive
  const int win_lo = streaming ? (int)(blockIdx.x * (16 * PWAVES)) + off - local + 1 : 0;  // first local key of the WG's first row
  uint8_t* vt = vtile + wave * PVTILE;
  const int tr_off = (4 * l4 + (l15 >> 2)) * PVROW + (l15 & 3) * 8;
  const int vtok = lane >> 1, vhalf = lane & 1;                  // V staging: 2 lanes per key row, 128 B each

  for (int kb = 0; kb < k_hi; kb += 32) {
    if (streaming && kb >= sink && kb + 32 <= win_lo) continue;  // tile entirely in the masked band (uniform)
    // ---- S^T tile: 2 groups of 16 keys --------------------------------------------------------------
    v4f st[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int key = kb + 16 * u + l15;
      const int kc = key < len_k ? key : (len_k - 1);
      const half_t* kp = p.k + (size_t)(k_begin + kc) * p.k_stride + (size_t)hk * PDH + 8 * l4;
      v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const v8h a = *reinterpret_cast<const v8h*>(kp + 32 * s);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qb[s], acc, 0, 0, 0);
      }
      st[u] = acc;   // st[u][r] = K[kb + 16u + 4*l4 + r] . Q[qrow]
    }
    // ---- stage V tile [32 keys][128 dims] fp16 into LDS (per wave) -----------------------------------
    {
      const int key = kb + vtok;
      const int kc = key < len_k ? key : (len_k - 1);
      const half_t* vp = p.v + (size_t)(k_begin + kc) * p.v_stride + (size_t)hk * PDH + vhalf * 64;
      uint8_t* dst = vt + vtok * PVROW + vhalf * 128;
#pragma unroll
      for (int w = 0; w < 8; ++w)
        *reinterpret_cast<v8h*>(dst + w * 16) = *reinterpret_cast<const v8h*>(vp + w * 8);
    }
    // ---- online softmax for query row `qrow` over this lane's 8 keys ----------------------------------
    float sv[8];
    float tmax = -1e30f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kb + 16 * u + 4 * l4 + r;
        bool ok = key < len_k && qrow < len_q;
        if (p.causal) ok = ok && key <= qrow + off;
        if (streaming) ok = ok && (key < sink || (qrow + off) - key < local);
        const float x = ok ? st[u][r] * scale : -1e30f;
        sv[u * 4 + r] = x;
        tmax = __builtin_fmaxf(tmax, x);
      }
    tmax = __builtin_fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = __builtin_fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = __builtin_fmaxf(m_run, tmax);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.0f;
    v8h pb;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float pe = sv[e] > -1e29f ? __expf(sv[e] - m_new) : 0.0f;
      const half_t ph = (half_t)pe;
      pb[e] = ph;
      psum += (float)ph;
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int c = 0; c < 8; ++c) oacc[c] *= alpha;
    // ---- O^T += V^T P^T ------------------------------------------------------------------------------------
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint8_t* src = vt + tr_off + c * 32;
      const pv4hp lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
          (__attribute__((address_space(3))) pv4hp*)(__attribute__((address_space(3))) void*)(src));
      const pv4hp hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
          (__attribute__((address_space(3))) pv4hp*)(__attribute__((address_space(3))) void*)(src + 16 * PVROW));
      const v8h a = {(half_t)lo[0], (half_t)lo[1], (half_t)lo[2], (half_t)lo[3],
                     (half_t)hi[0], (half_t)hi[1], (half_t)hi[2], (half_t)hi[3]};
      oacc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb, oacc[c], 0, 0, 0);
    }
  }
  // ---- finish: row sum over the 4 lanes of a query row, normalise, store ------------------------------------
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  if (qrow >= len_q) return;
  const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
  half_t* op = p.out + ((size_t)(q_begin + qrow) * p.num_heads + h) * PDH + 4 * l4;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    typedef _Float16 v4h_t __attribute__((ext_vector_type(4)));
    const v4h_t o = {(half_t)(oacc[c][0] * inv), (half_t)(oacc[c][1] * inv), (half_t)(oacc[c][2] * inv),
                     (half_t)(oacc[c][3] * inv)};
    *reinterpret_cast<v4h_t*>(op + c * 16) = o;
  }
}

}  // namespace omni

using namespace omni;

extern "C" int omni_prefill_attention(void* out_f16, const void* q_f16, const void* k_f16, const void* v_f16,
                                      int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                      const void* cu_seqlens_q_i32, const void* cu_seqlens_k_i32, int batch,
                                      int max_seqlen_q, int num_heads, int num_kv_heads, int head_dim, int causal,
                                      const void* head_mask_type_i32, const void* streaming_info_i32, void* stream) {
  if (!out_f16 || !q_f16 || !k_f16 || !v_f16 || !cu_seqlens_q_i32 || !cu_seqlens_k_i32) return OMNI_EINVAL;
  if (head_dim != PDH || batch < 1 || num_heads < 1 || num_kv_heads < 1 || num_heads % num_kv_heads != 0 ||
      max_seqlen_q < 1 || q_stride % 8 != 0 || k_stride % 8 != 0 || v_stride % 8 != 0)
    return OMNI_EINVAL;
  if ((head_mask_type_i32 == nullptr) != (streaming_info_i32 == nullptr)) return OMNI_EINVAL;
  PrefillArgs a;
  a.q = (const half_t*)q_f16; a.k = (const half_t*)k_f16; a.v = (const half_t*)v_f16; a.out = (half_t*)out_f16;
  a.q_stride = q_stride; a.k_stride = k_stride; a.v_stride = v_stride;
  a.cu_q = (const int*)cu_seqlens_q_i32; a.cu_k = (const int*)cu_seqlens_k_i32;
  a.head_mask_type = (const int*)head_mask_type_i32; a.streaming_info = (const int*)streaming_info_i32;
  a.num_heads = num_heads; a.num_kv_heads = num_kv_heads; a.causal = causal;
  dim3 grid((max_seqlen_q + 16 * PWAVES - 1) / (16 * PWAVES), num_heads, batch);
  hipLaunchKernelGGL(prefill_attn_kernel, grid, dim3(64 * PWAVES), 0, (hipStream_t)stream, a);
  return omni_launch_status();
}
