"""Mirror of omniserve_backend.fused_attention_fine_grained_dense
(kernels/csrc/fused_attention/fused_attention_fine_grained/: fine_grained_common/update_kv_cache.h:16-43,
dense_attention/fused_attention.h:18-46): KV4 cache with retrieval + streaming heads (LServe), and the
all-retrieval configuration QServe prefill uses (ctx_update_kv.py:96-135)."""
from ._attn_common import (compute_padding_offsets, decode_attention_fine_grained,  # noqa: F401
                           prefill_write, prefill_write_fine_grained)


def apply_bias_rope_update_kv_cache(qkv, retrieval_seq_lens, streaming_seq_lens, padding_offset,
                                    retrieval_kv_pointers, streaming_kv_pointers, retrieval_head_flags,
                                    head_rank_table, head_num, kv_head_num, seq_len, tokens_per_block,
                                    size_per_retrieval_token, size_per_streaming_token, sink_token_num,
                                    local_token_num, sink_block_num, local_block_num,
                                    num_retrieval_kv_heads, num_streaming_kv_heads, rotary_embedding_dim,
                                    rotary_embedding_base, rotary_embedding_scale,
                                    rotary_embedding_max_positions, neox_rotary_style, int4_kv_cache,
                                    kv_cache_with_zeros):
    """Prefill: RoPE q,k in place + KV4 quantise/write.  The rotary scale type is LINEAR: the angle uses
    pos / rotary_embedding_scale (fine_grained_common/update_kv_cache.cu:75, applyBiasRopeUpdateKVCache.h:596).
    `streaming_seq_lens` is unused by the reference kernel as well (both classes use the real length)."""
    what = "fused_attention_fine_grained_dense.apply_bias_rope_update_kv_cache"
    if num_streaming_kv_heads == 0 and num_retrieval_kv_heads == kv_head_num:
        prefill_write(qkv, retrieval_seq_lens, padding_offset, retrieval_kv_pointers, head_num, kv_head_num,
                      seq_len, tokens_per_block, size_per_retrieval_token, rotary_embedding_dim,
                      rotary_embedding_base, 1.0 / float(rotary_embedding_scale), rotary_embedding_max_positions,
                      neox_rotary_style, int4_kv_cache, kv_cache_with_zeros, what)
        return
    prefill_write_fine_grained(qkv, retrieval_seq_lens, padding_offset, retrieval_kv_pointers, streaming_kv_pointers,
                               retrieval_head_flags, head_rank_table, head_num, kv_head_num, seq_len,
                               tokens_per_block, size_per_retrieval_token, size_per_streaming_token, sink_token_num,
                               local_token_num, sink_block_num, local_block_num, num_retrieval_kv_heads,
                               num_streaming_kv_heads, rotary_embedding_dim, rotary_embedding_base,
                               1.0 / float(rotary_embedding_scale), rotary_embedding_max_positions,
                               neox_rotary_style, int4_kv_cache, kv_cache_with_zeros, what)


def single_query_attention(q, k, v, retrieval_kv_pointers, streaming_kv_pointers, retrieval_head_flags,
                           head_rank_table, length_per_sample_, alibi_slopes_, memory_max_seqlen, tokens_per_block,
                           size_per_retrieval_token, size_per_streaming_token, sink_token_num, local_token_num,
                           sink_block_num, local_block_num, num_retrieval_kv_heads, num_streaming_kv_heads,
                           timestep, rotary_embedding_dim, rotary_base, rotary_embedding_scale, neox_rotary_style,
                           int4_kv_cache, kv_cache_with_zeros, multiblock_switch):
    """Decode attention with retrieval + streaming heads (decoding_attention.py:326-353).  Returns a new
    fp16 [B,Hq,Dh] tensor.  `multiblock_switch` (the reference's split-KV threshold) is accepted and ignored:
    the KV split is planned from the problem size."""
    if alibi_slopes_ is not None:
        raise NotImplementedError("alibi slopes are not used by the QServe/LServe Llama path")
    if length_per_sample_ is None:
        raise NotImplementedError("length_per_sample is required")
    ts = min(int(timestep), int(memory_max_seqlen)) if memory_max_seqlen else int(timestep)
    return decode_attention_fine_grained(
        q, k, v, retrieval_kv_pointers, streaming_kv_pointers, retrieval_head_flags, head_rank_table, None,
        length_per_sample_, tokens_per_block, size_per_retrieval_token, size_per_streaming_token, sink_token_num,
        local_token_num, sink_block_num, local_block_num, num_retrieval_kv_heads, num_streaming_kv_heads, ts,
        rotary_embedding_dim, rotary_base, 1.0 / float(rotary_embedding_scale), neox_rotary_style, int4_kv_cache,
        kv_cache_with_zeros, 0, "fused_attention_fine_grained_dense.single_query_attention")
