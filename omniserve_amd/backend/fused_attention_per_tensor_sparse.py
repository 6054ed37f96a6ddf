"""Mirror of omniserve_backend.fused_attention_per_tensor_sparse
(kernels/csrc/fused_attention/fused_attention_per_tensor/sparse_attention/fused_attention.h:18-50):
per-tensor KV8 decode attention where each retrieval q head attends only its dynamically selected pages."""
from ._attn_common import decode_attention_fine_grained


def single_query_attention(q, k, v, kv_scale_quant_orig_, kv_scale_orig_quant_, retrieval_kv_pointers,
                           streaming_kv_pointers, retrieval_head_flags, head_rank_table, dynamic_sparse_page_idxes_,
                           length_per_sample_, alibi_slopes_, memory_max_seqlen, tokens_per_block,
                           size_per_retrieval_token, size_per_streaming_token, sink_token_num, local_token_num,
                           sink_block_num, local_block_num, num_retrieval_kv_heads, num_streaming_kv_heads, timestep,
                           rotary_embedding_dim, rotary_base, rotary_embedding_scale, neox_rotary_style,
                           int4_kv_cache, kv_cache_with_zeros, tokens_per_sub_chunk, hidden_dim_per_retrieval_token,
                           multiblock_switch):
    """decoding_attention.py:239-306.  dynamic_sparse_page_idxes_ int32 [B,Hq,pages] (None: all pages); the K pages
    carry min/max statistics, which the appended key updates."""
    what = "fused_attention_per_tensor_sparse.single_query_attention"
    if alibi_slopes_ is not None:
        raise NotImplementedError("alibi slopes are not used by the QServe/LServe Llama path")
    if length_per_sample_ is None:
        raise NotImplementedError("length_per_sample is required")
    if hidden_dim_per_retrieval_token != num_retrieval_kv_heads * q.shape[-1]:
        raise RuntimeError(what + ": hidden_dim_per_retrieval_token != num_retrieval_kv_heads * head_dim")
    ts = min(int(timestep), int(memory_max_seqlen)) if memory_max_seqlen else int(timestep)
    return decode_attention_fine_grained(
        q, k, v, retrieval_kv_pointers, streaming_kv_pointers, retrieval_head_flags, head_rank_table,
        dynamic_sparse_page_idxes_, length_per_sample_, tokens_per_block, size_per_retrieval_token,
        size_per_streaming_token, sink_token_num, local_token_num, sink_block_num, local_block_num,
        num_retrieval_kv_heads, num_streaming_kv_heads, ts, rotary_embedding_dim, rotary_base,
        1.0 / float(rotary_embedding_scale), neox_rotary_style, int4_kv_cache, kv_cache_with_zeros,
        tokens_per_sub_chunk, what, kv_scale_quant_orig=kv_scale_quant_orig_,
        kv_scale_orig_quant=kv_scale_orig_quant_, per_tensor=True)
