"""Mirror of omniserve_backend.fused_attention_pure_dense
(kernels/csrc/fused_attention/fused_attention_pure_dense/fused_attention.cpp:150-256)."""
from ._attn_common import compute_padding_offsets, decode_attention, prefill_write  # noqa: F401


def single_query_attention(q, k, v, kv_pointers, length_per_sample_, alibi_slopes_, memory_max_seqlen,
                           tokens_per_block, size_per_token, timestep, rotary_embedding_dim,
                           rotary_base, neox_rotary_style, int4_kv_cache, kv_cache_with_zeros):
    """Decode attention over the KV4 paged cache; RoPE + KV append fused.  Returns a new fp16
    [B,Hq,Dh] tensor (callee allocates, as the reference).  `timestep` (max context length in the
    batch, decoding_attention.py:153) sizes the KV split; lengths include the current token."""
    if alibi_slopes_ is not None:
        raise NotImplementedError("alibi slopes are not used by the QServe/LServe Llama path")
    if length_per_sample_ is None:
        raise NotImplementedError("length_per_sample is required")
    return decode_attention(q, k, v, kv_pointers, length_per_sample_, tokens_per_block, size_per_token,
                            min(int(timestep), int(memory_max_seqlen)) if memory_max_seqlen else timestep,
                            rotary_embedding_dim, rotary_base, neox_rotary_style, int4_kv_cache,
                            kv_cache_with_zeros, "fused_attention_pure_dense.single_query_attention")


def apply_bias_rope_update_kv_cache(qkv, seq_lens, padding_offset, kv_pointers, head_num, kv_head_num,
                                    seq_len, tokens_per_block, size_per_token, rotary_embedding_dim,
                                    rotary_embedding_base, rotary_embedding_max_positions,
                                    neox_rotary_style, int4_kv_cache, kv_cache_with_zeros):
    """Prefill: RoPE q,k in place + KV4 quantise/write (update_kv_cache.h:11-27)."""
    prefill_write(qkv, seq_lens, padding_offset, kv_pointers, head_num, kv_head_num, seq_len,
                  tokens_per_block, size_per_token, rotary_embedding_dim, rotary_embedding_base, 1.0,
                  rotary_embedding_max_positions, neox_rotary_style, int4_kv_cache, kv_cache_with_zeros,
                  "fused_attention_pure_dense.apply_bias_rope_update_kv_cache")
