"""Mirror of omniserve_backend.fused_attention_per_tensor_dense
(kernels/csrc/fused_attention/fused_attention_per_tensor/: per_tensor_common/update_kv_cache.h:16-44,
dense_attention/fused_attention.h:18-46): the per-tensor KV8 cache of LServe's published configuration
(`w8a8kv8`, `--kv-quant-granularity per_tensor`) with retrieval + streaming heads."""
from ._attn_common import decode_attention_fine_grained, prefill_write_fine_grained


def apply_bias_rope_update_kv_cache(qkv, kv_scale_orig_quant, retrieval_seq_lens, streaming_seq_lens, padding_offset,
                                    retrieval_kv_pointers, streaming_kv_pointers, retrieval_head_flags,
                                    head_rank_table, head_num, kv_head_num, seq_len, tokens_per_block,
                                    size_per_retrieval_token, size_per_streaming_token, sink_token_num,
                                    local_token_num, sink_block_num, local_block_num,
                                    num_retrieval_kv_heads, num_streaming_kv_heads, rotary_embedding_dim,
                                    rotary_embedding_base, rotary_embedding_scale,
                                    rotary_embedding_max_positions, neox_rotary_style, int4_kv_cache,
                                    kv_cache_with_zeros):
    """Prefill (ctx_update_kv.py:49-92): RoPE q,k in place + int8 quantise/write with the static scales
    kv_scale_orig_quant fp32 [2] (K, V).  LINEAR rotary scaling: angle = pos / rotary_embedding_scale."""
    prefill_write_fine_grained(qkv, retrieval_seq_lens, padding_offset, retrieval_kv_pointers, streaming_kv_pointers,
                               retrieval_head_flags, head_rank_table, head_num, kv_head_num, seq_len,
                               tokens_per_block, size_per_retrieval_token, size_per_streaming_token, sink_token_num,
                               local_token_num, sink_block_num, local_block_num, num_retrieval_kv_heads,
                               num_streaming_kv_heads, rotary_embedding_dim, rotary_embedding_base,
                               1.0 / float(rotary_embedding_scale), rotary_embedding_max_positions,
                               neox_rotary_style, int4_kv_cache, kv_cache_with_zeros,
                               "fused_attention_per_tensor_dense.apply_bias_rope_update_kv_cache",
                               kv_scale_orig_quant=kv_scale_orig_quant, per_tensor=True)


def single_query_attention(q, k, v, kv_scale_quant_orig_, kv_scale_orig_quant_, retrieval_kv_pointers,
                           streaming_kv_pointers, retrieval_head_flags, head_rank_table, length_per_sample_,
                           alibi_slopes_, memory_max_seqlen, tokens_per_block, size_per_retrieval_token,
                           size_per_streaming_token, sink_token_num, local_token_num, sink_block_num,
                           local_block_num, num_retrieval_kv_heads, num_streaming_kv_heads, timestep,
                           rotary_embedding_dim, rotary_base, rotary_embedding_scale, neox_rotary_style,
                           int4_kv_cache, kv_cache_with_zeros, multiblock_switch):
    """Decode attention on per-tensor KV8 pages (decoding_attention.py:185-236).  Returns a new fp16 [B,Hq,Dh]
    tensor; `multiblock_switch` is accepted and ignored (the KV split is planned from the problem size)."""
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
        kv_cache_with_zeros, 0, "fused_attention_per_tensor_dense.single_query_attention",
        kv_scale_quant_orig=kv_scale_quant_orig_, kv_scale_orig_quant=kv_scale_orig_quant_, per_tensor=True)
