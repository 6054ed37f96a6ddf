"""Oracle (CPU, numpy) for prefill attention.  Test infrastructure only.

The arithmetic lives in the un-vendored, un-pinned third-party package `block_sparse_attn`
(github mit-han-lab/Block-Sparse-Attention; absent from /root/reference and from pyproject.toml).
This restates its published semantics -- FlashAttention-2 varlen causal attention, and the
"token streaming" Lambda mask of DuoAttention/LServe (sink + local tokens; "block streaming": the same rule on 128-token
blocks, ctx_attn_func.py:47-59, dead code upstream) -- anchored on the
reference's call sites (omniserve/modeling/layers/ctx_attn/ctx_attn_func.py:39-45,68-73,
ctx_attn_init.py:28-50).  PARITY UNPINNED (no reference tests or golden vectors exist at that boundary).
"""
import numpy as np


def streaming_mask(qi, ki, sink, local, block=1):
    """Keys a streaming head's query may see besides causality.  block = 1: token_streaming_attn_func (the first `sink` tokens
    and the last `local` tokens, the query's own included); block = 128: block_streaming_attn_func, the same rule on 128-token
    block indices (the first `sink` key blocks, the query's own block and the local - 1 blocks before it).  qi: bottom-right
    aligned query positions [lq, 1], ki: key positions [1, lk]."""
    return ((ki // block) < sink) | (((qi // block) - (ki // block)) < local)


def varlen_attention(q, k, v, cu_q, cu_k, causal=True, head_mask_type=None, streaming_info=None, block=1):
    """q [Lq,Hq,D], k,v [Lk,Hkv,D] fp16 -> out fp16 [Lq,Hq,D].  f64 softmax reference.  block: granularity of streaming_info
    (streaming_mask)."""
    q = np.asarray(q, np.float16); k = np.asarray(k, np.float16); v = np.asarray(v, np.float16)
    Lq, Hq, D = q.shape
    Hk = k.shape[1]
    g = Hq // Hk
    out = np.zeros((Lq, Hq, D), np.float16)
    scale = 1.0 / np.sqrt(D)
    for b in range(len(cu_q) - 1):
        q0, q1, k0, k1 = int(cu_q[b]), int(cu_q[b + 1]), int(cu_k[b]), int(cu_k[b + 1])
        lq, lk = q1 - q0, k1 - k0
        if lq == 0:
            continue
        off = lk - lq
        qi = np.arange(lq)[:, None] + off
        ki = np.arange(lk)[None, :]
        for h in range(Hq):
            s = (q[q0:q1, h].astype(np.float64) @ k[k0:k1, h // g].astype(np.float64).T) * scale
            mask = np.ones((lq, lk), bool)
            if causal:
                mask &= ki <= qi
            if head_mask_type is not None and int(head_mask_type[h]) < 0:
                sink, local = int(streaming_info[2 * h]), int(streaming_info[2 * h + 1])
                mask &= streaming_mask(qi, ki, sink, local, block)
            s = np.where(mask, s, -np.inf)
            m = s.max(axis=1, keepdims=True)
            p = np.exp(s - m)
            p = np.where(mask, p, 0.0)
            o = (p @ v[k0:k1, h // g].astype(np.float64)) / p.sum(axis=1, keepdims=True)
            out[q0:q1, h] = o.astype(np.float16)
    return out


def varlen_attention_rows(q, k, v, cu_q, cu_k, rows, causal=True, head_mask_type=None, streaming_info=None,
                          heads=None, chunk=256):
    """The same reference restricted to the global query rows `rows` (sorted int array) and the q heads `heads`
    (default: all) -> fp16 [len(rows), len(heads), D].  Evaluated chunk by chunk over the query rows and, for a
    Lambda-masked head, only over the keys the mask lets through, so that long sequences (L = 16K ... 64K) stay
    within seconds and a few hundred MB of host memory: the tests compare a dense sample of rows of a long case."""
    q = np.asarray(q, np.float16); k = np.asarray(k, np.float16); v = np.asarray(v, np.float16)
    rows = np.asarray(rows, np.int64)
    Lq, Hq, D = q.shape
    g = Hq // k.shape[1]
    heads = list(range(Hq)) if heads is None else list(heads)
    out = np.zeros((len(rows), len(heads), D), np.float16)
    scale = 1.0 / np.sqrt(D)
    seq_of = np.searchsorted(np.asarray(cu_q, np.int64), rows, side="right") - 1
    for b in np.unique(seq_of):
        q0, q1, k0, k1 = int(cu_q[b]), int(cu_q[b + 1]), int(cu_k[b]), int(cu_k[b + 1])
        off = (k1 - k0) - (q1 - q0)
        sel = np.nonzero(seq_of == b)[0]
        for c0 in range(0, len(sel), chunk):
            idx = sel[c0:c0 + chunk]
            qi = (rows[idx] - q0 + off)[:, None]                     # key position a query may look back from
            hi = int(qi.max()) + 1 if causal else (k1 - k0)
            for hn, h in enumerate(heads):
                streaming = head_mask_type is not None and int(head_mask_type[h]) < 0
                if streaming:
                    sink, local = int(streaming_info[2 * h]), int(streaming_info[2 * h + 1])
                    lo = max(min(sink, hi), int(qi.min()) - local + 1)
                    ki = np.concatenate([np.arange(0, min(sink, hi)), np.arange(max(lo, 0), hi)])
                    ki = np.unique(ki)
                else:
                    ki = np.arange(0, hi)
                kk = k[k0 + ki, h // g].astype(np.float64)
                s = (q[rows[idx], h].astype(np.float64) @ kk.T) * scale
                mask = np.ones(s.shape, bool)
                if causal:
                    mask &= ki[None, :] <= qi
                if streaming:
                    mask &= (ki[None, :] < sink) | ((qi - ki[None, :]) < local)
                s = np.where(mask, s, -np.inf)
                m = s.max(axis=1, keepdims=True)
                pr = np.where(mask, np.exp(s - m), 0.0)
                o = (pr @ v[k0 + ki, h // g].astype(np.float64)) / pr.sum(axis=1, keepdims=True)
                out[idx, hn] = o.astype(np.float16)
    return out
