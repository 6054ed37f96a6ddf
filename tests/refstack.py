"""Test harness: run the REFERENCE's own Python layers (omniserve/modeling/layers/*.py from /root/reference) on top of
this repo's `omniserve_backend` mirror -- on the CPU, without a GPU.

The reference modules import `omniserve_backend.*` / `block_sparse_attn`, which resolve to our mirror packages; the
mirror marshals tensors into the C ABI (raw pointers, sizes, strides).  Here the ctypes handle of libomniserve_hip.so
is swapped for `OracleLib`, an object with the same `omni_*` entry points that interprets the raw pointers as HOST
memory and computes with oracle/ (numpy).  What this pins, against the reference's real call sites and argument
values: module / function names, positional order and meaning of every argument, output ownership (caller-allocated
buffers written in place vs returned tensors), strides of the fused-qkv views, the raw-pointer block tables.  It is
test infrastructure (the oracle is the arithmetic), never a product path: the product still fails without the HIP
library and rejects host tensors (tests/test_cabi_cpu.py)."""
import contextlib
import ctypes
import os
import sys

import numpy as np
import torch

from oracle import attention as oattn
from oracle import elementwise as oe
from oracle import kv4, kv8, w4a8

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F16 = np.float16


def reference_available():
    return os.path.isdir(os.path.join(REF, "omniserve"))


def _arr(ptr, shape, dtype):
    """numpy view of host memory at raw address `ptr`."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if n == 0:
        return np.zeros(shape, dtype)
    buf = (ctypes.c_uint8 * n).from_address(int(ptr))
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def _rows(ptr, rows, cols, stride, dtype):
    """[rows, cols] view of a row-strided matrix (stride in elements)."""
    item = np.dtype(dtype).itemsize
    if rows == 0:
        return np.zeros((0, cols), dtype)
    n = ((rows - 1) * stride + cols) * item
    buf = (ctypes.c_uint8 * n).from_address(int(ptr))
    flat = np.frombuffer(buf, dtype=dtype)
    return np.lib.stride_tricks.as_strided(flat, shape=(rows, cols), strides=(stride * item, item))


class _Pages:
    """The pages a [B,2,max_blocks] raw-pointer block table references, gathered into oracle PagedKV4 pools."""

    def __init__(self, table_ptr, B, max_blocks, needed_blocks, heads, D, tpb):
        self.tab = _arr(table_ptr, (B, 2, max_blocks), np.int64)
        self.page_bytes = kv4.page_bytes(heads, D, tpb)
        n = int(sum(needed_blocks))
        self.k, self.v = kv4.PagedKV4(max(n, 1), heads, D, tpb), kv4.PagedKV4(max(n, 1), heads, D, tpb)
        self.kt = np.zeros((B, max_blocks), np.int64)
        self.vt = np.zeros((B, max_blocks), np.int64)
        self.map = []
        nxt = 0
        for b in range(B):
            for j in range(int(needed_blocks[b])):
                for which, cache, t in ((0, self.k, self.kt), (1, self.v, self.vt)):
                    src = _arr(self.tab[b, which, j], (self.page_bytes,), np.uint8)
                    cache.pool[nxt, : self.page_bytes] = src
                    t[b, j] = nxt
                    self.map.append((cache, nxt, src))
                nxt += 1

    def scatter(self):
        for cache, idx, dst in self.map:
            dst[:] = cache.pool[idx, : self.page_bytes]


class _FGPages:
    """Both page pools of one layer (retrieval + streaming ring) referenced by two raw-pointer tables, gathered into an
    oracle FineGrainedKV.  stats_sub > 0: the retrieval K pages carry min/max statistics behind data + tail
    (cache_engine.py:84-131) and those bytes travel too; otherwise only data + tail are touched."""

    def __init__(self, rp, sp, B, rb, sb, need_r, nr, ns, D, tpb, stats_sub, flags, rank, sink, local, sink_blocks,
                 local_blocks, kv8_scales=None):
        self.map = []

        def make(heads, n, which, stats):
            if kv8_scales is None:
                return kv4.PagedKV4(max(n, 1), max(heads, 1), D, tpb, stats_sub_chunk=stats)
            return kv8.PagedKV8(max(n, 1), max(heads, 1), D, kv8_scales[which], tpb, stats_sub_chunk=stats)

        def gather(ptr, blocks, heads, need, stats):
            n = int(sum(need))
            k, v = make(heads, n, 0, stats), make(heads, n, 1, 0)
            kt, vt = np.zeros((B, max(blocks, 1)), np.int64), np.zeros((B, max(blocks, 1)), np.int64)
            if heads == 0 or not ptr:
                return k, v, kt, vt
            tab = _arr(ptr, (B, 2, blocks), np.int64)
            nxt = 0
            for b in range(B):
                for j in range(int(need[b])):
                    for which, cache, t in ((0, k, kt), (1, v, vt)):
                        src = _arr(tab[b, which, j], (cache.page_bytes,), np.uint8)
                        cache.pool[nxt] = src
                        t[b, j] = nxt
                        self.map.append((cache, nxt, src))
                    nxt += 1
            return k, v, kt, vt

        rk, rv, rkt, rvt = gather(rp, rb, nr, need_r, stats_sub)
        sk, sv, skt, svt = gather(sp, sb, ns, [sb] * B, 0)
        self.fg = kv4.FineGrainedKV(rk, rv, rkt, rvt, sk, sv, skt, svt, flags, rank, sink, local, sink_blocks,
                                    local_blocks, stats_sub)
        self.rk, self.rkt = rk, rkt

    def scatter(self):
        for cache, idx, dst in self.map:
            dst[:] = cache.pool[idx]


class OracleLib:
    """Same entry points as libomniserve_hip.so (include/omniserve_hip.h), host memory, oracle arithmetic."""

    def __init__(self):
        self.calls = []
        self.rope = (10000.0, 1.0)       # (base, scale) of the last rope_table() the mirror built

    # ---- introspection / sizing ------------------------------------------------------------------------------
    def omni_abi_version(self):
        return 4

    def omni_gemm_workspace_bytes(self, M, N, K):
        return 1 << 16

    def omni_gemm_partial_workspace_bytes(self, M, N, K):
        return 1 << 16

    def omni_kv4_decode_workspace_bytes(self, B, H, D, ctx):
        return 1 << 16

    # ---- GEMMs ---------------------------------------------------------------------------------------------------
    def omni_w4a8_per_chn_gemm(self, a, qw, wscales, ascales, wsz, asum, out, M, N, K, stride, ws, wsb, stream):
        self.calls.append("omni_w4a8_per_chn_gemm")
        res = w4a8.gemm_per_chn(_arr(a, (M, K), np.int8), _arr(qw, (N, K // 2), np.int8), _arr(wscales, (N,), F16),
                                _arr(ascales, (M,), F16), _arr(wsz, (N,), F16), _arr(asum, (M,), F16))
        _rows(out, M, N, stride, F16)[:] = res
        return 0

    def omni_w4a8_per_group_gemm(self, a, qw, zeros, scales_i8, wscales, ascales, out, M, N, K, stride, ws, wsb, stream):
        self.calls.append("omni_w4a8_per_group_gemm")
        res = w4a8.gemm_per_group(_arr(a, (M, K), np.int8), _arr(qw, (N, K // 2), np.int8),
                                  _arr(zeros, (K // 128, N), np.int8), _arr(scales_i8, (K // 128, N), np.int8),
                                  _arr(wscales, (N,), F16), _arr(ascales, (M,), F16))
        _rows(out, M, N, stride, F16)[:] = res
        return 0

    def omni_w8a8_gemm(self, a, w, wscales, ascales, out, M, N, K, stride, ws, wsb, stream):
        self.calls.append("omni_w8a8_gemm")
        res = w4a8.gemm_w8a8(_arr(a, (M, K), np.int8), _arr(w, (N, K), np.int8), _arr(wscales, (N,), F16),
                             _arr(ascales, (M,), F16))
        _rows(out, M, N, stride, F16)[:] = res
        return 0

    # ---- row kernels -------------------------------------------------------------------------------------------------
    def omni_quant(self, out, x, scale, tokens, hidden, stream):
        self.calls.append("omni_quant")
        q, s, _ = oe.quant_per_token(_arr(x, (tokens, hidden), F16), fuse_sum=False)
        _arr(out, (tokens, hidden), np.int8)[:] = q
        _arr(scale, (tokens,), F16)[:] = s
        return 0

    def omni_quant_fuse_sum(self, out, x, sm, scale, tokens, hidden, stream):
        self.calls.append("omni_quant_fuse_sum")
        q, s, t = oe.quant_per_token(_arr(x, (tokens, hidden), F16), fuse_sum=True)
        _arr(out, (tokens, hidden), np.int8)[:] = q
        _arr(scale, (tokens,), F16)[:] = s
        _arr(sm, (tokens,), F16)[:] = t
        return 0

    def omni_rms_norm(self, out, x, w, eps, tokens, hidden, stream):
        self.calls.append("omni_rms_norm")
        _arr(out, (tokens, hidden), F16)[:] = oe.rms_norm(_arr(x, (tokens, hidden), F16), _arr(w, (hidden,), F16), eps)
        return 0

    def omni_rms_norm_general(self, out, x, w, scale, eps, tokens, hidden, stream):
        self.calls.append("omni_rms_norm_general")
        q, s, _ = oe.rms_norm_general(_arr(x, (tokens, hidden), F16), _arr(w, (hidden,), F16), eps, False)
        _arr(out, (tokens, hidden), np.int8)[:] = q
        _arr(scale, (tokens,), F16)[:] = s
        return 0

    def omni_rms_norm_general_fuse_sum(self, out, x, w, sm, scale, eps, tokens, hidden, stream):
        self.calls.append("omni_rms_norm_general_fuse_sum")
        q, s, t = oe.rms_norm_general(_arr(x, (tokens, hidden), F16), _arr(w, (hidden,), F16), eps, True)
        _arr(out, (tokens, hidden), np.int8)[:] = q
        _arr(scale, (tokens,), F16)[:] = s
        _arr(sm, (tokens,), F16)[:] = t
        return 0

    def omni_silu_and_mul(self, out, x, tokens, d, stream):
        self.calls.append("omni_silu_and_mul")
        _arr(out, (tokens, d), F16)[:] = oe.silu_and_mul(_arr(x, (tokens, 2 * d), F16))
        return 0

    # ---- overloads off the Llama path (csrc/offpath.hip) ---------------------------------------------------------------
    def omni_quant_static(self, out, x, scale, tokens, hidden, stream):
        self.calls.append("omni_quant_static")
        _arr(out, (tokens, hidden), np.int8)[:] = oe.quant_static(_arr(x, (tokens, hidden), F16), scale)
        return 0

    def omni_dequant(self, out, x, scale, tokens, hidden, in_stride, out_stride, stream):
        self.calls.append("omni_dequant")
        _rows(out, tokens, hidden, out_stride, F16)[:] = oe.dequant(_rows(x, tokens, hidden, in_stride, np.int32), scale)
        return 0

    def omni_dequant_add_residual(self, out, x, res, tok_scale, scale, tokens, hidden, stream):
        self.calls.append("omni_dequant_add_residual")
        sc = _arr(tok_scale, (tokens,), F16) if tok_scale else scale
        _arr(out, (tokens, hidden), F16)[:] = oe.dequant_add_residual(_arr(x, (tokens, hidden), np.int32),
                                                                      _arr(res, (tokens, hidden), F16), sc)
        return 0

    def omni_rms_norm_quant(self, out, x, w, eps, tokens, hidden, stream):
        self.calls.append("omni_rms_norm_quant")
        _arr(out, (tokens, hidden), np.int8)[:] = oe.rms_norm_quant(_arr(x, (tokens, hidden), F16), _arr(w, (hidden,), F16), eps)
        return 0

    def omni_rms_norm_general_static(self, out, x, w, scaling, eps, tokens, hidden, stream):
        self.calls.append("omni_rms_norm_general_static")
        _arr(out, (tokens, hidden), np.int8)[:] = oe.rms_norm_general_static(
            _arr(x, (tokens, hidden), F16), _arr(w, (hidden,), F16), _arr(scaling, (1,), F16), eps)
        return 0

    def omni_dequant_add_residual_rms_norm_quant(self, out, x, res, gamma, tok_scale, scale, eps, tokens, hidden, stream):
        self.calls.append("omni_dequant_add_residual_rms_norm_quant")
        sc = _arr(tok_scale, (tokens,), F16) if tok_scale else scale
        r = _arr(res, (tokens, hidden), F16)
        q, nr = oe.dequant_add_residual_rms_norm_quant(_arr(x, (tokens, hidden), np.int32), r.copy(),
                                                       _arr(gamma, (hidden,), F16), sc, eps)
        _arr(out, (tokens, hidden), np.int8)[:] = q
        r[:] = nr
        return 0

    def omni_gelu(self, out, x, kind, tokens, d, stream):
        self.calls.append("omni_gelu")
        _arr(out, (tokens, d), F16)[:] = (oe.gelu_fast if kind else oe.gelu_new)(_arr(x, (tokens, d), F16))
        return 0

    def omni_dequant_silu_and_mul_quant(self, out, x, sg, su, so, tok_scale, tmp, tokens, d, stream):
        self.calls.append("omni_dequant_silu_and_mul_quant")
        acc = _arr(x, (tokens, 2 * d), np.int32)
        if tok_scale:
            q, s, t = oe.dequant_silu_and_mul_quant(acc, sg, su)
            _arr(tok_scale, (tokens,), np.float32)[:] = s
            _arr(tmp, (tokens, d), np.float32)[:] = t
        else:
            q = oe.dequant_silu_and_mul_quant(acc, sg, su, so)
        _arr(out, (tokens, d), np.int8)[:] = q
        return 0

    # ---- KV4 cache -------------------------------------------------------------------------------------------------------
    def omni_compute_padding_offsets(self, out, cu, batch, max_len, total, stream):
        self.calls.append("omni_compute_padding_offsets")
        _arr(out, (total,), np.int32)[:] = kv4.compute_padding_offsets(_arr(cu, (batch + 1,), np.int32), max_len)
        return 0

    def omni_kv4_prefill_write(self, qkv, seq_lens, pad, kv_pointers, tokens, batch, max_blocks, Hq, Hkv, D, max_seq,
                               tpb, rope, rope_max, max_pos, stream):
        self.calls.append("omni_kv4_prefill_write")
        lens = _arr(seq_lens, (batch,), np.int32)
        pg = _Pages(kv_pointers, batch, max_blocks, [(int(n) + tpb - 1) // tpb for n in lens], Hkv, D, tpb)
        x = _arr(qkv, (tokens, (Hq + 2 * Hkv) * D), F16)
        base, scale = self.rope
        x[:] = kv4.prefill_write(x, lens, pg.k, pg.v, pg.kt, pg.vt, Hq, Hkv, D, base, 1.0 / scale)
        pg.scatter()
        return 0

    def omni_kv4_decode_attention(self, out, q, k, v, q_stride, kv_stride, kv_pointers, lengths, B, max_blocks, Hq, Hkv,
                                  D, tpb, max_ctx, rope, rope_max, ws, wsb, stream):
        self.calls.append("omni_kv4_decode_attention")
        lens = _arr(lengths, (B,), np.int32)
        pg = _Pages(kv_pointers, B, max_blocks, [(int(n) - 1) // tpb + 1 for n in lens], Hkv, D, tpb)
        qa = _rows(q, B, Hq * D, q_stride, F16).reshape(B, Hq, D)
        ka = _rows(k, B, Hkv * D, kv_stride, F16).reshape(B, Hkv, D)
        va = _rows(v, B, Hkv * D, kv_stride, F16).reshape(B, Hkv, D)
        res = kv4.decode_attention(qa, ka, va, lens, pg.k, pg.v, pg.kt, pg.vt, self.rope[0])
        _arr(out, (B, Hq, D), F16)[:] = res
        pg.scatter()
        return 0

    # ---- LServe: fine-grained KV4 / per-tensor KV8 pools with streaming rings, statistics pooling, page selector ------------
    def _fg_prefill(self, qkv, seq_lens, rp, sp, flags, rank, tokens, batch, rb, sb, Hq, Hkv, nr, ns, D, tpb, sink, local,
                    sink_blocks, local_blocks, kv8_scales):
        lens = _arr(seq_lens, (batch,), np.int32)
        fl, rk = _arr(flags, (Hkv,), np.int32), _arr(rank, (Hkv,), np.int32)
        pg = _FGPages(rp, sp, batch, rb, sb, [(int(n) + tpb - 1) // tpb for n in lens], nr, ns, D, tpb, 0, fl, rk, sink,
                      local, sink_blocks, local_blocks, kv8_scales)
        x = _arr(qkv, (tokens, (Hq + 2 * Hkv) * D), F16)
        base, factor = self.rope
        x[:] = kv4.prefill_write_fine_grained(x, lens, pg.fg, Hq, Hkv, D, base, factor)
        pg.scatter()
        return 0

    def omni_kv4_prefill_write_fine_grained(self, qkv, seq_lens, pad, rp, sp, flags, rank, tokens, batch, rb, sb, Hq, Hkv,
                                            nr, ns, D, max_seq, tpb, sink, local, sink_blocks, local_blocks, rope, rope_len,
                                            max_pos, stream):
        self.calls.append("omni_kv4_prefill_write_fine_grained")
        return self._fg_prefill(qkv, seq_lens, rp, sp, flags, rank, tokens, batch, rb, sb, Hq, Hkv, nr, ns, D, tpb, sink,
                                local, sink_blocks, local_blocks, None)

    def omni_kv8_prefill_write_per_tensor(self, qkv, kv_oq, seq_lens, pad, rp, sp, flags, rank, tokens, batch, rb, sb, Hq,
                                          Hkv, nr, ns, D, max_seq, tpb, sink, local, sink_blocks, local_blocks, rope,
                                          rope_len, max_pos, stream):
        self.calls.append("omni_kv8_prefill_write_per_tensor")
        oq = _arr(kv_oq, (2,), np.float32)
        return self._fg_prefill(qkv, seq_lens, rp, sp, flags, rank, tokens, batch, rb, sb, Hq, Hkv, nr, ns, D, tpb, sink,
                                local, sink_blocks, local_blocks, tuple((np.float32(1.0) / oq).astype(np.float32)))

    def _fg_decode(self, out, q, k, v, q_stride, kv_stride, rp, sp, flags, rank, lengths, dyn_ptr, ndyn, sub, B, rb, sb, Hq,
                   Hkv, nr, ns, D, tpb, sink, local, sink_blocks, local_blocks, kv8_scales):
        lens = _arr(lengths, (B,), np.int32)
        fl, rk = _arr(flags, (Hkv,), np.int32), _arr(rank, (Hkv,), np.int32)
        dyn = _arr(dyn_ptr, (B, Hq, ndyn), np.int32) if dyn_ptr else None
        pg = _FGPages(rp, sp, B, rb, sb, [(int(n) - 1) // tpb + 1 for n in lens], nr, ns, D, tpb, sub if dyn is not None else 0,
                      fl, rk, sink, local, sink_blocks, local_blocks, kv8_scales)
        qa = _rows(q, B, Hq * D, q_stride, F16).reshape(B, Hq, D)
        ka = _rows(k, B, Hkv * D, kv_stride, F16).reshape(B, Hkv, D)
        va = _rows(v, B, Hkv * D, kv_stride, F16).reshape(B, Hkv, D)
        base, factor = self.rope
        _arr(out, (B, Hq, D), F16)[:] = kv4.decode_attention_fine_grained(qa, ka, va, lens, pg.fg, base, factor, dyn)
        pg.scatter()
        return 0

    def omni_kv4_decode_attention_fine_grained(self, out, q, k, v, q_stride, kv_stride, rp, sp, flags, rank, lengths, dyn_ptr,
                                               ndyn, sub, B, rb, sb, Hq, Hkv, nr, ns, D, tpb, sink, local, sink_blocks,
                                               local_blocks, max_ctx, rope, rope_len, ws, wsb, stream):
        self.calls.append("omni_kv4_decode_attention_fine_grained")
        return self._fg_decode(out, q, k, v, q_stride, kv_stride, rp, sp, flags, rank, lengths, dyn_ptr, ndyn, sub, B, rb, sb,
                               Hq, Hkv, nr, ns, D, tpb, sink, local, sink_blocks, local_blocks, None)

    def omni_kv8_decode_attention_per_tensor(self, out, q, k, v, q_stride, kv_stride, kv_qo, kv_oq, rp, sp, flags, rank,
                                             lengths, dyn_ptr, ndyn, sub, B, rb, sb, Hq, Hkv, nr, ns, D, tpb, sink, local,
                                             sink_blocks, local_blocks, max_ctx, rope, rope_len, ws, wsb, stream):
        self.calls.append("omni_kv8_decode_attention_per_tensor")
        qo = _arr(kv_qo, (2,), np.float32)
        return self._fg_decode(out, q, k, v, q_stride, kv_stride, rp, sp, flags, rank, lengths, dyn_ptr, ndyn, sub, B, rb, sb,
                               Hq, Hkv, nr, ns, D, tpb, sink, local, sink_blocks, local_blocks, (qo[0], qo[1]))

    def _retrieval_k_pool(self, kv_ptrs, batch, nblk, need, heads, D, row_bytes, tpb, sub):
        """K pages (with statistics) of the retrieval pool as an oracle pool + index table; returns (pages, scatter)."""
        kv8_scales = None if row_bytes == D // 2 else (1.0, 1.0)
        pg = _FGPages(kv_ptrs, 0, batch, nblk, 0, need, heads, 0, D, tpb, sub, [1] * heads, list(range(heads)), 0, 0, 0, 0,
                      kv8_scales)
        return pg

    def omni_kv_min_max_pool(self, x, kv_ptrs, cu_seqlens, heads_idx, batch, nblk, H_in, pool_h, D, row_bytes, max_seqlen,
                             pooling_size, page_size, stream):
        self.calls.append("omni_kv_min_max_pool")
        cu = _arr(cu_seqlens, (batch + 1,), np.int32)
        need = [(int(cu[b + 1] - cu[b]) + page_size - 1) // page_size for b in range(batch)]
        pg = self._retrieval_k_pool(kv_ptrs, batch, nblk, need, pool_h, D, row_bytes, page_size, pooling_size)
        keys = _arr(x, (int(cu[-1]), H_in, D), F16)
        kv4.paged_min_max_pool(keys, cu, [int(h) for h in _arr(heads_idx, (pool_h,), np.int32)], pg.rk.pool, pg.rkt,
                               page_size, pooling_size, row_bytes=row_bytes)
        pg.scatter()
        return 0

    def omni_kv_page_selector(self, out, q, q_stride, kv_ptrs, flags, rank, lengths, B, nblk, Hq, Hkv, nr, D, row_bytes, tpb,
                              sub, padded, rope, rope_len, stream):
        self.calls.append("omni_kv_page_selector")
        lens = _arr(lengths, (B,), np.int32)
        need = [(int(n) - 1 + tpb - 1) // tpb for n in lens]        # pages that hold history tokens
        pg = self._retrieval_k_pool(kv_ptrs, B, nblk, need, nr, D, row_bytes, tpb, sub)
        qa = _rows(q, B, Hq * D, q_stride, F16).reshape(B, Hq, D)
        base, scale = self.rope
        want = kv4.page_selector(qa, lens, _arr(flags, (Hkv,), np.int32), _arr(rank, (Hkv,), np.int32), pg.rk.pool, pg.rkt,
                                 Hkv, nr, tpb, sub, base, scale, row_bytes=row_bytes)
        o = _arr(out, (B, Hq, padded), F16)          # zero-initialised by the mirror; sized on `timestep` by the caller
        o[:, :, : want.shape[2]] = want[:, :, : padded]
        return 0

    def omni_prefill_attention(self, out, q, k, v, qs, ks, vs, cu_q, cu_k, batch, max_q, Hq, Hkv, D, causal, hm, si, stream):
        self.calls.append("omni_prefill_attention")
        cq, ck = _arr(cu_q, (batch + 1,), np.int32), _arr(cu_k, (batch + 1,), np.int32)
        Lq, Lk = int(cq[-1]), int(ck[-1])
        qa = _rows(q, Lq, Hq * D, qs, F16).reshape(Lq, Hq, D)
        ka = _rows(k, Lk, Hkv * D, ks, F16).reshape(Lk, Hkv, D)
        va = _rows(v, Lk, Hkv * D, vs, F16).reshape(Lk, Hkv, D)
        hma = _arr(hm, (Hq,), np.int32) if hm else None
        sia = _arr(si, (2 * Hq,), np.int32) if si else None
        _arr(out, (Lq, Hq, D), F16)[:] = oattn.varlen_attention(qa, ka, va, cq, ck, bool(causal), hma, sia)
        return 0


@contextlib.contextmanager
def reference_over_mirror():
    """Context: `import omniserve...` resolves on this CPU-only box, the mirror's C-ABI handle is an OracleLib."""
    from omniserve_amd import _lib, rope as rope_mod
    from omniserve_amd.backend import _attn_common
    for p in (REF, ROOT):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    fake = OracleLib()
    saved = (_lib._lib, _lib.require_cuda, _lib.current_stream, _attn_common.rope_table, torch.cuda.current_device,
             torch.Tensor.cuda)
    saved_factories = {n: getattr(torch, n) for n in ("zeros", "empty", "ones", "tensor", "full")}

    def on_host(fn):      # the reference's layers allocate with device="cuda" (w4a8_linear.py:44-100): host memory here
        def wrapped(*a, **k):
            if isinstance(k.get("device"), int) or str(k.get("device", "")).startswith("cuda"):
                k["device"] = "cpu"
            return fn(*a, **k)
        return wrapped
    real_rope = rope_mod.rope_table

    def recording_rope(max_pos, dim, base, scale, device):
        fake.rope = (float(base), float(scale))
        return real_rope(max_pos, dim, base, scale, device)

    _lib._lib = fake
    _lib.require_cuda = lambda *t: None
    _lib.current_stream = lambda: 0
    _attn_common.rope_table = recording_rope
    if not torch.cuda.is_available():
        torch.cuda.current_device = lambda: 0
        torch.Tensor.cuda = lambda self, *a, **k: self        # from_linear moves the weight to the GPU (w4a8_linear.py:286)
        for n, fn in saved_factories.items():
            setattr(torch, n, on_host(fn))
    try:
        yield fake
    finally:
        for n, fn in saved_factories.items():
            setattr(torch, n, fn)
        (_lib._lib, _lib.require_cuda, _lib.current_stream, _attn_common.rope_table, torch.cuda.current_device,
         torch.Tensor.cuda) = saved
