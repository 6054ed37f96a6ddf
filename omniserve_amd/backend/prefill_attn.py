"""Host side of the prefill-attention boundary: the three functions the reference imports from the
un-vendored `block_sparse_attn` package (ctx_attn_func.py:3-7) plus the `flash_attn` import shim."""
import torch

from .. import _lib


def _run(q, k, v, cu_q, cu_k, max_q, causal, head_mask_type, streaming_info, what, block=False):
    _lib.require_cuda(q, k, v, cu_q, cu_k)
    if q.dtype != torch.float16 or q.dim() != 3 or q.shape[-1] != 128:
        raise RuntimeError("%s: q must be fp16 [tokens, heads, 128]" % what)
    for t in (q, k, v):
        if t.stride(2) != 1 or t.stride(1) != 128:
            raise RuntimeError("%s: heads must be contiguous (token-strided views are fine)" % what)
    if cu_q.dtype != torch.int32 or cu_k.dtype != torch.int32:
        raise RuntimeError("%s: cu_seqlens must be int32" % what)
    out = torch.empty((q.shape[0], q.shape[1], 128), dtype=torch.float16, device=q.device)
    hm = head_mask_type.data_ptr() if head_mask_type is not None else None
    si = streaming_info.data_ptr() if streaming_info is not None else None
    if block:
        rc = _lib.lib().omni_prefill_attention_block_streaming(
            out.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0), v.stride(0),
            cu_q.data_ptr(), cu_k.data_ptr(), cu_q.shape[0] - 1, int(max_q), q.shape[1], k.shape[1], 128,
            hm, si, _lib.current_stream())
    else:
        rc = _lib.lib().omni_prefill_attention(
            out.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0), v.stride(0),
            cu_q.data_ptr(), cu_k.data_ptr(), cu_q.shape[0] - 1, int(max_q), q.shape[1], k.shape[1], 128,
            1 if causal else 0, hm, si, _lib.current_stream())
    _lib.check(rc, what)
    return out


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0,
                           softmax_scale=None, causal=False, **unused):
    if dropout_p != 0.0:
        raise NotImplementedError("dropout is not used at inference")
    if softmax_scale is not None and abs(softmax_scale - 128 ** -0.5) > 1e-9:
        raise NotImplementedError("only the default softmax scale 1/sqrt(head_dim)")
    return _run(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, causal, None, None, "flash_attn_varlen_func")


def token_streaming_attn_func(q, k, v, cu_seqlens_q, cu_seqlens_k, head_mask_type, streaming_info, max_seqlen_q,
                              max_seqlen_k, **unused):
    """Causal attention where heads with head_mask_type < 0 see only sink + local tokens
    (streaming_info = [sink, local] per q head)."""
    if head_mask_type.dtype != torch.int32 or streaming_info.dtype != torch.int32:
        raise RuntimeError("token_streaming_attn_func: head_mask_type / streaming_info must be int32")
    return _run(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, True, head_mask_type, streaming_info,
                "token_streaming_attn_func")


def block_streaming_attn_func(q, k, v, cu_seqlens_q, cu_seqlens_k, head_mask_type, streaming_info, max_seqlen_q,
                              max_seqlen_k, p_dropout=0.0, **unused):
    """Causal attention where heads with head_mask_type < 0 see only sink + local BLOCKS of 128 tokens (streaming_info =
    [sink_blocks, local_blocks] per q head; the query's own block counts as one local block).  Imported by the reference
    (ctx_attn_func.py:3-7) and wrapped by block_static_sparse_attn (:47-59), which nothing calls."""
    if p_dropout != 0.0:
        raise NotImplementedError("dropout is not used at inference")
    if head_mask_type.dtype != torch.int32 or streaming_info.dtype != torch.int32:
        raise RuntimeError("block_streaming_attn_func: head_mask_type / streaming_info must be int32")
    return _run(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, True, head_mask_type, streaming_info,
                "block_streaming_attn_func", block=True)
