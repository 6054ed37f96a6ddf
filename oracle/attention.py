"""Oracle (CPU, numpy) for prefill attention.  Test infrastructure only.

The arithmetic lives in the un-vendored, un-pinned third-party package `block_sparse_attn`
(github mit-han-lab/Block-Sparse-Attention; absent from /root/reference and from pyproject.toml).
This restates its published semantics -- FlashAttention-2 varlen causal attention, and the
"token streaming" Lambda mask of DuoAttention/LServe (sink + local tokens) -- anchored on the
reference's call sites (omniserve/modeling/layers/ctx_attn/ctx_attn_func.py:39-45,68-73,
ctx_attn_init.py:28-50).  PARITY UNPINNED (no reference tests or golden vectors exist at that boundary).
"""
import numpy as np


def varlen_attention(q, k, v, cu_q, cu_k, causal=True, head_mask_type=None, streaming_info=None):
    """q [Lq,Hq,D], k,v [Lk,Hkv,D] fp16 -> out fp16 [Lq,Hq,D].  f64 softmax reference."""
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
                mask &= (ki < sink) | ((qi - ki) < local)
            s = np.where(mask, s, -np.inf)
            m = s.max(axis=1, keepdims=True)
            p = np.exp(s - m)
            p = np.where(mask, p, 0.0)
            o = (p @ v[k0:k1, h // g].astype(np.float64)) / p.sum(axis=1, keepdims=True)
            out[q0:q1, h] = o.astype(np.float16)
    return out
