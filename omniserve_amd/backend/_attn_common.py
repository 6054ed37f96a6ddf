from __future__ import annotations

import torch

from .. import _lib
from ..rope import rope_table


def _check_cfg(rotary_embedding_dim, neox_rotary_style, int4_kv_cache, kv_cache_with_zeros, head_dim):
    if int4_kv_cache and not kv_cache_with_zeros:
        # upstream writes SIGNED 4-bit codes for this format (cast_to_packed_int4, decoderMaskedMultiheadAttentionUtils.h:
        # 1890-1904) and reads them back UNSIGNED (float_from_int4: u & 0x0F, :1642-1647): no working behaviour to match
        raise NotImplementedError("KV4 without zero points (int4_kv_cache, kv_cache_with_zeros=False) is not implemented: the "
                                  "reference decodes its signed codes as unsigned (DESIGN.md section 7); use the fine_grained "
                                  "(zeros) format")
    if not (int4_kv_cache and kv_cache_with_zeros):
        raise NotImplementedError("only the KV4 + zeros (fine_grained) cache format is implemented")
    if not neox_rotary_style:
        raise NotImplementedError("only neox-style RoPE is implemented")
    if head_dim != 128 or rotary_embedding_dim != head_dim:
        raise NotImplementedError("head_dim = rotary_dim = 128 only")


def _check_cfg_kv8(rotary_embedding_dim, neox_rotary_style, int4_kv_cache, kv_cache_with_zeros, head_dim):
    """per_tensor family: int8 pages with static scales and no zero points (arg_utils.py:492-503)."""
    if int4_kv_cache or kv_cache_with_zeros:
        raise NotImplementedError("per_tensor: only the KV8 format without zero points is implemented")
    if not neox_rotary_style:
        raise NotImplementedError("only neox-style RoPE is implemented")
    if head_dim != 128 or rotary_embedding_dim != head_dim:
        raise NotImplementedError("head_dim = rotary_dim = 128 only")


def _kv_scale(t, what, name):
    if t is None or t.dtype != torch.float32 or t.numel() < 2 or not t.is_contiguous():
        raise RuntimeError("%s: %s must be a contiguous fp32 tensor with 2 elements (K, V)" % (what, name))
    _lib.require_cuda(t)
    return t


def compute_padding_offsets(cu_seqlens, max_seqlen, tot_num_tokens):
    """-> int32 [tot_num_tokens], out[tok] = b*max_seqlen - cu_seqlens[b]
    (common/input_metadata_helper.cu:38-50; callee allocates)."""
    _lib.require_cuda(cu_seqlens)
    out = torch.empty((tot_num_tokens,), dtype=torch.int32, device=cu_seqlens.device)
    rc = _lib.lib().omni_compute_padding_offsets(out.data_ptr(), cu_seqlens.data_ptr(),
                                                 cu_seqlens.shape[0] - 1, int(max_seqlen),
                                                 int(tot_num_tokens), _lib.current_stream())
    _lib.check(rc, "compute_padding_offsets")
    return out


def prefill_write(qkv, seq_lens, padding_offset, kv_pointers, head_num, kv_head_num, seq_len,
                  tokens_per_block, size_per_token, rotary_embedding_dim, rotary_embedding_base,
                  rope_scale, rotary_embedding_max_positions, neox, int4, zeros, what):
    _lib.require_cuda(qkv, seq_lens, padding_offset, kv_pointers)
    head_dim = qkv.shape[-1] // (head_num + 2 * kv_head_num)
    _check_cfg(rotary_embedding_dim, neox, int4, zeros, head_dim)
    if size_per_token != kv_head_num * head_dim // 2:
        raise RuntimeError("%s: size_per_token %d != Hkv*Dh/2" % (what, size_per_token))
    if not qkv.is_contiguous() or qkv.dtype != torch.float16:
        raise RuntimeError("%s: qkv must be contiguous fp16" % what)
    table = rope_table(int(seq_len), head_dim, float(rotary_embedding_base), float(rope_scale), qkv.device)
    rc = _lib.lib().omni_kv4_prefill_write(
        qkv.data_ptr(), seq_lens.data_ptr(), padding_offset.data_ptr(), kv_pointers.data_ptr(),
        qkv.shape[0], seq_lens.shape[0], kv_pointers.shape[-1], head_num, kv_head_num, head_dim,
        int(seq_len), int(tokens_per_block), table.data_ptr(), table.shape[0],
        int(rotary_embedding_max_positions), _lib.current_stream())
    _lib.check(rc, what)


_dense_ws_bytes = {}      # (B, Hq, D, context rounded up to 1024) -> omni_kv4_decode_workspace_bytes (monotone in the context)


def decode_attention(q, k, v, kv_pointers, lengths, tokens_per_block, size_per_token, timestep,
                     rotary_embedding_dim, rotary_base, neox, int4, zeros, what):
    # (runs once per decoder layer of every eager decode step: attributes are read once, the scratch size is cached)
    f = _lib.fast()
    if f is not None:      # the pybind11 body (csrc_ext/omni_ext.cpp): same checks, same C-ABI call, callee-allocated result
        B, Hq, D = q.shape
        _check_cfg(rotary_embedding_dim, neox, int4, zeros, D)
        max_ctx = max(int(timestep), 1)
        device = q.device
        table = rope_table(max_ctx + 1, D, float(rotary_base), 1.0, device)
        key = (B, Hq, D, (max_ctx + 1023) >> 10)
        need = _dense_ws_bytes.get(key)
        if need is None:
            need = _dense_ws_bytes[key] = int(_lib.lib().omni_kv4_decode_workspace_bytes(B, Hq, D, key[3] << 10))
        return f.decode_attention_kv4(q, k, v, kv_pointers, lengths, int(tokens_per_block), int(size_per_token), max_ctx, table,
                                      _lib.workspace(need, device, "attn"), what)
    _lib.require_cuda(q, k, v, kv_pointers, lengths)
    B, Hq, D = q.shape
    Hkv = k.shape[1]
    _check_cfg(rotary_embedding_dim, neox, int4, zeros, D)
    if size_per_token != Hkv * D // 2:
        raise RuntimeError("%s: size_per_token %d != Hkv*Dh/2" % (what, size_per_token))
    qs, ks, vs = q.stride(), k.stride(), v.stride()
    if q.dtype is not torch.float16 or qs[2] != 1 or qs[1] != D:
        raise RuntimeError("%s: q must be fp16 [B,H,D] with contiguous heads" % what)
    if ks[2] != 1 or ks[1] != D or vs[2] != 1 or vs[1] != D:
        raise RuntimeError("%s: k/v heads must be contiguous" % what)      # TORCH_CHECK in the reference
    if ks[0] != vs[0]:
        raise RuntimeError("%s: k and v must share the row stride" % what)
    if lengths.dtype is not torch.int32 or not kv_pointers.is_contiguous():
        raise RuntimeError("%s: lengths must be int32, kv_pointers contiguous" % what)
    max_ctx = max(int(timestep), 1)
    device = q.device
    table = rope_table(max_ctx + 1, D, float(rotary_base), 1.0, device)
    key = (B, Hq, D, (max_ctx + 1023) >> 10)
    need = _dense_ws_bytes.get(key)
    if need is None:
        need = _dense_ws_bytes[key] = int(_lib.lib().omni_kv4_decode_workspace_bytes(B, Hq, D, key[3] << 10))
    ws = _lib.workspace(need, device, "attn")
    out = torch.empty((B, Hq, D), dtype=torch.float16, device=device)
    rc = _lib.lib().omni_kv4_decode_attention(
        out.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), qs[0], ks[0],
        kv_pointers.data_ptr(), lengths.data_ptr(), B, kv_pointers.shape[-1], Hq, Hkv, D,
        int(tokens_per_block), max_ctx, table.data_ptr(), table.shape[0], ws.data_ptr(), ws.numel(),
        _lib.current_stream())
    _lib.check(rc, what)
    return out


def _fg_tables(retrieval_kv_pointers, streaming_kv_pointers, num_retrieval_kv_heads, num_streaming_kv_heads, what):
    if num_retrieval_kv_heads > 0 and (retrieval_kv_pointers is None or not retrieval_kv_pointers.is_contiguous()):
        raise RuntimeError("%s: retrieval_kv_pointers must be a contiguous int64 [B,2,blocks] tensor" % what)
    if num_streaming_kv_heads > 0 and (streaming_kv_pointers is None or not streaming_kv_pointers.is_contiguous()):
        raise RuntimeError("%s: streaming_kv_pointers must be a contiguous int64 [B,2,blocks] tensor" % what)
    rp = retrieval_kv_pointers.data_ptr() if num_retrieval_kv_heads > 0 else 0
    rb = retrieval_kv_pointers.shape[-1] if num_retrieval_kv_heads > 0 else 0
    sp = streaming_kv_pointers.data_ptr() if num_streaming_kv_heads > 0 else 0
    sb = streaming_kv_pointers.shape[-1] if num_streaming_kv_heads > 0 else 0
    return rp, rb, sp, sb


def prefill_write_fine_grained(qkv, seq_lens, padding_offset, retrieval_kv_pointers, streaming_kv_pointers,
                               retrieval_head_flags, head_rank_table, head_num, kv_head_num, seq_len,
                               tokens_per_block, size_per_retrieval_token, size_per_streaming_token, sink_token_num,
                               local_token_num, sink_block_num, local_block_num, num_retrieval_kv_heads,
                               num_streaming_kv_heads, rotary_embedding_dim, rotary_embedding_base, rope_scale,
                               rotary_embedding_max_positions, neox, int4, zeros, what, kv_scale_orig_quant=None,
                               per_tensor=False):
    """per_tensor=True: the KV8 family (kv_scale_orig_quant fp32 [2] on the device)."""
    _lib.require_cuda(qkv, seq_lens, padding_offset, retrieval_head_flags, head_rank_table)
    head_dim = qkv.shape[-1] // (head_num + 2 * kv_head_num)
    if per_tensor:
        _check_cfg_kv8(rotary_embedding_dim, neox, int4, zeros, head_dim)
        kv_scale_orig_quant = _kv_scale(kv_scale_orig_quant, what, "kv_scale_orig_quant")
        row = head_dim
    else:
        _check_cfg(rotary_embedding_dim, neox, int4, zeros, head_dim)
        row = head_dim // 2
    if size_per_retrieval_token != num_retrieval_kv_heads * row or \
            size_per_streaming_token != num_streaming_kv_heads * row:
        raise RuntimeError("%s: size_per_*_token must be heads_in_pool*%d" % (what, row))
    if not qkv.is_contiguous() or qkv.dtype != torch.float16:
        raise RuntimeError("%s: qkv must be contiguous fp16" % what)
    if retrieval_head_flags.dtype != torch.int32 or head_rank_table.dtype != torch.int32:
        raise RuntimeError("%s: retrieval_head_flags / head_rank_table must be int32" % what)
    rp, rb, sp, sb = _fg_tables(retrieval_kv_pointers, streaming_kv_pointers, num_retrieval_kv_heads,
                                num_streaming_kv_heads, what)
    table = rope_table(int(seq_len), head_dim, float(rotary_embedding_base), float(rope_scale), qkv.device)
    if per_tensor:
        rc = _lib.lib().omni_kv8_prefill_write_per_tensor(
            qkv.data_ptr(), kv_scale_orig_quant.data_ptr(), seq_lens.data_ptr(), padding_offset.data_ptr(), rp, sp,
            retrieval_head_flags.data_ptr(), head_rank_table.data_ptr(), qkv.shape[0], seq_lens.shape[0], rb, sb,
            head_num, kv_head_num, int(num_retrieval_kv_heads), int(num_streaming_kv_heads), head_dim, int(seq_len),
            int(tokens_per_block), int(sink_token_num), int(local_token_num), int(sink_block_num),
            int(local_block_num), table.data_ptr(), table.shape[0], int(rotary_embedding_max_positions),
            _lib.current_stream())
        _lib.check(rc, what)
        return
    rc = _lib.lib().omni_kv4_prefill_write_fine_grained(
        qkv.data_ptr(), seq_lens.data_ptr(), padding_offset.data_ptr(), rp, sp, retrieval_head_flags.data_ptr(),
        head_rank_table.data_ptr(), qkv.shape[0], seq_lens.shape[0], rb, sb, head_num, kv_head_num,
        int(num_retrieval_kv_heads), int(num_streaming_kv_heads), head_dim, int(seq_len), int(tokens_per_block),
        int(sink_token_num), int(local_token_num), int(sink_block_num), int(local_block_num), table.data_ptr(),
        table.shape[0], int(rotary_embedding_max_positions), _lib.current_stream())
    _lib.check(rc, what)


def decode_attention_fine_grained(q, k, v, retrieval_kv_pointers, streaming_kv_pointers, retrieval_head_flags,
                                  head_rank_table, dynamic_sparse_page_idxes, lengths, tokens_per_block,
                                  size_per_retrieval_token, size_per_streaming_token, sink_token_num, local_token_num,
                                  sink_block_num, local_block_num, num_retrieval_kv_heads, num_streaming_kv_heads,
                                  timestep, rotary_embedding_dim, rotary_base, rope_scale, neox, int4, zeros,
                                  tokens_per_sub_chunk, what, kv_scale_quant_orig=None, kv_scale_orig_quant=None,
                                  per_tensor=False, merge_quant=None, merge_f16=None):
    """per_tensor=True: the KV8 family (both scale tensors fp32 [2] on the device).
    merge_quant = (out_i8 [B, Hq*D], input_sum fp16 [B], scale fp16 [B]): fused extension -- the flash-decoding merge is
    done by the per-token quantiser that follows upstream (invoke_quant[_fuse_sum]); returns None.
    merge_f16 = (out_f16 [B, Hq*D], amax slots): fused extension -- the merge as the wide kernel that also raises the row
    maxima of |out| (omni_attn_merge_f16_amax); returns None."""
    _lib.require_cuda(q, k, v, lengths, retrieval_head_flags, head_rank_table)
    B, Hq, D = q.shape
    Hkv = k.shape[1]
    if per_tensor:
        _check_cfg_kv8(rotary_embedding_dim, neox, int4, zeros, D)
        kv_scale_quant_orig = _kv_scale(kv_scale_quant_orig, what, "kv_scale_quant_orig")
        kv_scale_orig_quant = _kv_scale(kv_scale_orig_quant, what, "kv_scale_orig_quant")
        row = D
    else:
        _check_cfg(rotary_embedding_dim, neox, int4, zeros, D)
        row = D // 2
    if size_per_retrieval_token != num_retrieval_kv_heads * row or \
            size_per_streaming_token != num_streaming_kv_heads * row:
        raise RuntimeError("%s: size_per_*_token must be heads_in_pool*%d" % (what, row))
    if q.dtype != torch.float16 or q.stride(2) != 1 or q.stride(1) != D:
        raise RuntimeError("%s: q must be fp16 [B,H,D] with contiguous heads" % what)
    if k.stride(2) != 1 or k.stride(1) != D or v.stride(2) != 1 or v.stride(1) != D:
        raise RuntimeError("%s: k/v heads must be contiguous" % what)
    if k.stride(0) != v.stride(0):
        raise RuntimeError("%s: k and v must share the row stride" % what)
    if lengths.dtype != torch.int32 or retrieval_head_flags.dtype != torch.int32 or \
            head_rank_table.dtype != torch.int32:
        raise RuntimeError("%s: lengths / retrieval_head_flags / head_rank_table must be int32" % what)
    rp, rb, sp, sb = _fg_tables(retrieval_kv_pointers, streaming_kv_pointers, num_retrieval_kv_heads,
                                num_streaming_kv_heads, what)
    dyn_ptr, ndyn = 0, 0
    if dynamic_sparse_page_idxes is not None:
        d = dynamic_sparse_page_idxes
        if d.dtype != torch.int32 or not d.is_contiguous() or d.dim() != 3 or d.shape[0] != B or d.shape[1] != Hq:
            raise RuntimeError("%s: dynamic_sparse_page_idxes must be contiguous int32 [B,Hq,pages]" % what)
        dyn_ptr, ndyn = d.data_ptr(), d.shape[2]
    max_ctx = max(int(timestep), 1)
    table = rope_table(max_ctx + 1, D, float(rotary_base), float(rope_scale), q.device)
    # the sparse launch plans its KV splits on num_pages * tokens_per_block attended tokens, which may exceed a tiny
    # context: size the scratch for the larger of the two (the bound is monotone in the token count)
    need = _lib.lib().omni_kv4_decode_workspace_bytes(
        B, Hq, D, max(max_ctx, ndyn * int(tokens_per_block), int(sink_token_num) + int(local_token_num)))
    ws = _lib.workspace(need, q.device, "attn")
    if merge_quant is not None or merge_f16 is not None:
        import ctypes
        if merge_quant is not None:
            out_i8, input_sum, scale = merge_quant
            _lib.require_cuda(out_i8, input_sum, scale)
        else:
            out_f16, amax = merge_f16
            _lib.require_cuda(out_f16, amax)
            if out_f16.dtype != torch.float16 or out_f16.numel() != B * Hq * D or not out_f16.is_contiguous():
                raise RuntimeError("%s: out_f16 must be a contiguous fp16 [B, Hq*D] tensor" % what)
        ns = ctypes.c_int(0)
        rc = _lib.lib().omni_kv_decode_attention_fine_grained_partial(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0),
            kv_scale_quant_orig.data_ptr() if per_tensor else None, kv_scale_orig_quant.data_ptr() if per_tensor else None,
            rp, sp, retrieval_head_flags.data_ptr(), head_rank_table.data_ptr(), lengths.data_ptr(), dyn_ptr, ndyn,
            int(tokens_per_sub_chunk), B, rb, sb, Hq, Hkv, int(num_retrieval_kv_heads), int(num_streaming_kv_heads), D,
            int(tokens_per_block), int(sink_token_num), int(local_token_num), int(sink_block_num), int(local_block_num),
            max_ctx, table.data_ptr(), table.shape[0], ws.data_ptr(), ws.numel(), ctypes.byref(ns), _lib.current_stream())
        _lib.check(rc, what + " (partials)")
        ml_bytes = B * Hq * ns.value * 2 * 4
        if merge_f16 is not None:
            rc = _lib.lib().omni_attn_merge_f16_amax(out_f16.data_ptr(), ws.data_ptr(), ws.data_ptr() + ml_bytes, ns.value,
                                                     amax.data_ptr(), B, Hq, _lib.current_stream())
            _lib.check(rc, what + " (merge)")
            return None
        rc = _lib.lib().omni_attn_merge_quant_fuse_sum(out_i8.data_ptr(), ws.data_ptr(), ws.data_ptr() + ml_bytes, ns.value,
                                                       None if input_sum is None else input_sum.data_ptr(),
                                                       scale.data_ptr(), B, Hq, _lib.current_stream())
        _lib.check(rc, what + " (merge + quant)")
        return None
    out = torch.empty((B, Hq, D), dtype=q.dtype, device=q.device)
    if per_tensor:
        rc = _lib.lib().omni_kv8_decode_attention_per_tensor(
            out.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0),
            kv_scale_quant_orig.data_ptr(), kv_scale_orig_quant.data_ptr(), rp, sp,
            retrieval_head_flags.data_ptr(), head_rank_table.data_ptr(), lengths.data_ptr(), dyn_ptr, ndyn,
            int(tokens_per_sub_chunk), B, rb, sb, Hq, Hkv, int(num_retrieval_kv_heads), int(num_streaming_kv_heads),
            D, int(tokens_per_block), int(sink_token_num), int(local_token_num), int(sink_block_num),
            int(local_block_num), max_ctx, table.data_ptr(), table.shape[0], ws.data_ptr(), ws.numel(),
            _lib.current_stream())
        _lib.check(rc, what)
        return out
    rc = _lib.lib().omni_kv4_decode_attention_fine_grained(
        out.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0), rp, sp,
        retrieval_head_flags.data_ptr(), head_rank_table.data_ptr(), lengths.data_ptr(), dyn_ptr, ndyn,
        int(tokens_per_sub_chunk), B, rb, sb, Hq, Hkv, int(num_retrieval_kv_heads), int(num_streaming_kv_heads), D,
        int(tokens_per_block), int(sink_token_num), int(local_token_num), int(sink_block_num), int(local_block_num),
        max_ctx, table.data_ptr(), table.shape[0], ws.data_ptr(), ws.numel(), _lib.current_stream())
    _lib.check(rc, what)
    return out
