"""Mirror of omniserve_backend.fused_attention_selector (sparse_utils/KVPageSelector/fused_kv_page_selector.h:50-78)."""
import torch

from .. import _lib
from ..rope import rope_table


def single_query_page_selector(q, k, v, retrieval_kv_pointers, streaming_kv_pointers, retrieval_head_flags,
                               head_rank_table, dynamic_sparse_page_idxes_, length_per_sample_, alibi_slopes_,
                               memory_max_seqlen, tokens_per_block, size_per_retrieval_token, size_per_streaming_token,
                               sink_token_num, local_token_num, sink_block_num, local_block_num,
                               num_retrieval_kv_heads, num_streaming_kv_heads, timestep, rotary_embedding_dim,
                               rotary_base, rotary_embedding_scale, neox_rotary_style, int4_kv_cache,
                               kv_cache_with_zeros, tokens_per_sub_chunk, hidden_dim_per_retrieval_token,
                               multiblock_switch):
    """-> fp16 [B, Hq, padded_sub_chunks] page scores (zeros for streaming heads); callee allocates."""
    _lib.require_cuda(q, retrieval_kv_pointers, retrieval_head_flags, head_rank_table, length_per_sample_)
    if not (kv_cache_with_zeros and neox_rotary_style):
        # callers always pass kv_cache_with_zeros=True (decoding_attention.py:126): the statistics sit behind a
        # 4 B/token-head tail in every page format
        raise NotImplementedError("only the 4 B/token-head tail layout with neox RoPE is implemented")
    B, Hq, D = q.shape
    if num_retrieval_kv_heads < 1 or size_per_retrieval_token % num_retrieval_kv_heads != 0 or \
            size_per_retrieval_token // num_retrieval_kv_heads != (D // 2 if int4_kv_cache else D):
        raise RuntimeError("fused_attention_selector.single_query_page_selector: size_per_retrieval_token does not "
                           "match num_retrieval_kv_heads * head_dim%s" % ("/2" if int4_kv_cache else ""))
    row_bytes = size_per_retrieval_token // num_retrieval_kv_heads
    Hkv = k.shape[1]
    n_sub = (int(timestep) + tokens_per_sub_chunk - 1) // tokens_per_sub_chunk
    group = tokens_per_block // tokens_per_sub_chunk
    padded = (n_sub + group - 1) // group * group
    out = torch.zeros((B, Hq, padded), dtype=q.dtype, device=q.device)
    scale = 1.0 if rotary_embedding_scale == 1.0 else 1.0 / float(rotary_embedding_scale)
    table = rope_table(int(timestep) + 1, D, float(rotary_base), scale, q.device)
    rc = _lib.lib().omni_kv_page_selector(out.data_ptr(), q.data_ptr(), q.stride(0), retrieval_kv_pointers.data_ptr(),
                                          retrieval_head_flags.data_ptr(), head_rank_table.data_ptr(),
                                          length_per_sample_.data_ptr(), B, retrieval_kv_pointers.shape[-1], Hq, Hkv,
                                          int(num_retrieval_kv_heads), D, row_bytes, int(tokens_per_block),
                                          int(tokens_per_sub_chunk), padded, table.data_ptr(), table.shape[0],
                                          _lib.current_stream())
    _lib.check(rc, "fused_attention_selector.single_query_page_selector")
    return out
