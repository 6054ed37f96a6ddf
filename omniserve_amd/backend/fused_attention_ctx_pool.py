"""Mirror of omniserve_backend.fused_attention_ctx_pool (sparse_utils/ContextPool/context_pool_kernel.h:12-22)."""
import torch

from .. import _lib


def paged_min_max_pool(input, kv_ptrs, cu_seqlens, pooling_heads_idx, max_seqlen, pooling_size, page_size,
                       size_per_retrieval_token, kv_cache_with_zeros):
    """Per sub-chunk min/max of the (post-RoPE) keys into the K page tails; returns None."""
    _lib.require_cuda(input, kv_ptrs, cu_seqlens, pooling_heads_idx)
    if input.dtype != torch.float16 or not input.is_contiguous():
        raise RuntimeError("paged_min_max_pool: input must be contiguous fp16 [tokens, heads, head_dim]")
    if cu_seqlens.dtype != torch.int32 or pooling_heads_idx.dtype != torch.int32:
        raise RuntimeError("paged_min_max_pool: cu_seqlens / pooling_heads_idx must be int32")
    if not kv_cache_with_zeros:
        # the caller always passes True (ctx_update_kv.py:177): 4 B/token-head tail in every page format
        raise NotImplementedError("only the 4 B/token-head tail layout is implemented")
    pool_h = pooling_heads_idx.numel()
    D = input.shape[2]
    if pool_h == 0:
        return
    if size_per_retrieval_token not in (pool_h * D // 2, pool_h * D):   # KV4 or KV8 rows
        raise RuntimeError("paged_min_max_pool: size_per_retrieval_token does not match the pooled heads")
    row_bytes = size_per_retrieval_token // pool_h
    rc = _lib.lib().omni_kv_min_max_pool(input.data_ptr(), kv_ptrs.data_ptr(), cu_seqlens.data_ptr(),
                                         pooling_heads_idx.data_ptr(), cu_seqlens.numel() - 1, kv_ptrs.shape[-1],
                                         input.shape[1], pool_h, input.shape[2], row_bytes, int(max_seqlen), int(pooling_size),
                                         int(page_size), _lib.current_stream())
    _lib.check(rc, "fused_attention_ctx_pool.paged_min_max_pool")
