"""Mirror of omniserve_backend.fused_attention_fine_grained_dense
(kernels/csrc/fused_attention/fused_attention_fine_grained/: fine_grained_common/update_kv_cache.h:16-43,
dense_attention/fused_attention.h:18-46).  Round 1: the all-retrieval-heads (dense) configuration
that QServe uses (ctx_update_kv.py:96-135); streaming heads raise until the LServe rows land."""
from ._attn_common import compute_padding_offsets, prefill_write  # noqa: F401


def apply_bias_rope_update_kv_cache(qkv, retrieval_seq_lens, streaming_seq_lens, padding_offset,
                                    retrieval_kv_pointers, streaming_kv_pointers, retrieval_head_flags,
                                    head_rank_table, head_num, kv_head_num, seq_len, tokens_per_block,
                                    size_per_retrieval_token, size_per_streaming_token, sink_token_num,
                                    local_token_num, sink_block_num, local_block_num,
                                    num_retrieval_kv_heads, num_streaming_kv_heads, rotary_embedding_dim,
                                    rotary_embedding_base, rotary_embedding_scale,
                                    rotary_embedding_max_positions, neox_rotary_style, int4_kv_cache,
                                    kv_cache_with_zeros):
    if num_streaming_kv_heads != 0 or num_retrieval_kv_heads != kv_head_num:
        raise NotImplementedError("streaming (LServe) heads are not implemented yet")
    # rotary scale type is LINEAR: the angle uses pos / rotary_embedding_scale
    # (fine_grained_common/update_kv_cache.cu:75, applyBiasRopeUpdateKVCache.h:596)
    prefill_write(qkv, retrieval_seq_lens, padding_offset, retrieval_kv_pointers, head_num, kv_head_num,
                  seq_len, tokens_per_block, size_per_retrieval_token, rotary_embedding_dim,
                  rotary_embedding_base, 1.0 / float(rotary_embedding_scale), rotary_embedding_max_positions,
                  neox_rotary_style, int4_kv_cache, kv_cache_with_zeros,
                  "fused_attention_fine_grained_dense.apply_bias_rope_update_kv_cache")


def single_query_attention(*args, **kwargs):
    raise NotImplementedError("fine-grained (LServe) decode attention is not implemented yet; "
                              "QServe dense decode uses fused_attention_pure_dense")
