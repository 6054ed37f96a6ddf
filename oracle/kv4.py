"""Oracle (CPU, numpy) for the KV4 paged cache: prefill writer, decode
attention, padding offsets.  Test infrastructure only.

Restates (citations relative to the upstream checkout, kernels/csrc/fused_attention/):

* page layout        omniserve/worker/cache_engine.py:73-88,117-136 ; common/kvCacheUtils.h:53-164
* RoPE (neox)        common/decoderMaskedMultiheadAttentionUtils.h:1147-1165,2565-2620
* KV4 quantize/pack  fused_attention_pure_dense/decoderMaskedMultiheadAttentionUtils.h:1838-1884,2070-2077
* KV4 dequant        same file :2120-2213 (fp16 fma(u4, scale, -scale*zero))
* prefill writer     fused_attention_fine_grained/fine_grained_common/applyBiasRopeUpdateKVCache.h:101-503
* decode attention   fused_attention_pure_dense/decoderMaskedMultiheadAttentionTemplate.hpp:743-2222
* padding offsets    common/input_metadata_helper.cu:16-50

A K or V page of one layer is ``int4 data [H_kv][tokens_per_block][Dh/2 bytes]``
followed by ``fp16 scale [H_kv][tokens_per_block]`` and
``fp16 zero [H_kv][tokens_per_block]``; byte j of a token holds element 2j in
its low nibble and element 2j+1 in its high nibble.  In these oracles a page
pool is a numpy uint8 array [num_pages, page_bytes] and a block table holds
page *indices* (the product passes raw device pointers = base + idx*page_bytes,
as omniserve/utils/block_table_utils.py:62-93 builds them).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
F16 = np.float16


def page_bytes(num_kv_heads: int, head_dim: int, tokens_per_block: int = 64) -> int:
    return num_kv_heads * tokens_per_block * (head_dim // 2) + 2 * num_kv_heads * tokens_per_block * 2


def compute_padding_offsets(cu_seqlens: np.ndarray, max_len: int) -> np.ndarray:
    """pad_off[tok] = b*max_len - cu_seqlens[b] for tokens of sequence b."""
    out = np.zeros(int(cu_seqlens[-1]), np.int32)
    for b in range(len(cu_seqlens) - 1):
        out[cu_seqlens[b]:cu_seqlens[b + 1]] = b * max_len - cu_seqlens[b]
    return out


def rope_neox(x_h: np.ndarray, pos: np.ndarray, base: float, scale: float = 1.0) -> np.ndarray:
    """x fp16 [..., D], pos broadcastable to x.shape[:-1].  Pair (i, i+D/2):
    angle = (pos*scale) / base**(2i/D) in f32; out = h(cos*x - sin*y), h(cos*y + sin*x).
    """
    x = np.asarray(x_h, F16).astype(F32)
    D = x.shape[-1]
    half = D // 2
    i = np.arange(half, dtype=F32)
    denom = np.power(F32(base), (F32(2.0) * i / F32(D)).astype(F32)).astype(F32)
    t = (np.asarray(pos, F32)[..., None] * F32(scale)).astype(F32)
    ang = (t / denom).astype(F32)
    c = np.cos(ang).astype(F32)
    s = np.sin(ang).astype(F32)
    a, b = x[..., :half], x[..., half:]
    ra = ((c * a).astype(F32) - (s * b).astype(F32)).astype(F32)
    rb = ((c * b).astype(F32) + (s * a).astype(F32)).astype(F32)
    return np.concatenate([ra, rb], axis=-1).astype(F16)


def kv4_quant_params(x_h: np.ndarray):
    """Per (token, head) asymmetric params over the last axis:
    scale = h((max-min)/15), zero = h(-15*min/(max-min)); both re-read as fp16."""
    x = np.asarray(x_h, F16).astype(F32)
    mx = x.max(axis=-1)
    mn = x.min(axis=-1)
    rng = (mx - mn).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = (rng / F32(15.0)).astype(F32).astype(F16)
        zero = ((F32(-15.0) * mn).astype(F32) / rng).astype(F32).astype(F16)
    return scale, zero


def kv4_quantize(x_h, scale_h, zero_h) -> np.ndarray:
    """u4 = rni_sat_u8( x * (1/f32(scale)) + f32(zero) ) & 0xF  -> packed bytes [..., D/2]."""
    x = np.asarray(x_h, F16).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (F32(1.0) / np.asarray(scale_h, F16).astype(F32)).astype(F32)
    z = np.asarray(zero_h, F16).astype(F32)
    v = ((x * inv[..., None]).astype(F32) + z[..., None]).astype(F32)
    r = np.rint(v)
    r = np.where(np.isnan(r), 0.0, r)
    q = np.clip(r, 0, 255).astype(np.uint8) & 0xF
    return (q[..., 0::2] | (q[..., 1::2] << 4)).astype(np.uint8)


def kv4_dequant(packed: np.ndarray, scale_h, zero_h, fp16_math: bool = True) -> np.ndarray:
    """-> float32 values.  fp16_math: the reference's h(fma(h(u4), h(scale), h(-scale*zero)))
    (single fp16 rounding of the fma); else plain f32 (u4 - zero)*scale."""
    p = np.asarray(packed, np.uint8)
    u = np.empty(p.shape[:-1] + (p.shape[-1] * 2,), F32)
    u[..., 0::2] = p & 0xF
    u[..., 1::2] = p >> 4
    s = np.asarray(scale_h, F16).astype(F32)[..., None]
    z = np.asarray(zero_h, F16).astype(F32)[..., None]
    if fp16_math:
        c = (-(s) * z).astype(F32).astype(F16).astype(np.float64)
        # u (<=15) * s is exact in f64; one rounding to fp16 = fused fma
        return (u.astype(np.float64) * s.astype(np.float64) + c).astype(F16).astype(F32)
    return ((u - z) * s).astype(F32)


class PagedKV4:
    """Page-pool view helper (numpy).  pool: uint8 [num_pages, page_bytes]."""

    def __init__(self, num_pages, num_kv_heads, head_dim, tokens_per_block=64, fill=0, stats_sub_chunk=0):
        self.H, self.D, self.TPB = num_kv_heads, head_dim, tokens_per_block
        self.row_bytes = head_dim // 2
        self.bytes_per_seq = num_kv_heads * tokens_per_block * (head_dim // 2)
        self.page_bytes = page_bytes(num_kv_heads, head_dim, tokens_per_block)
        if stats_sub_chunk:   # K page with min/max statistics appended (cache_engine.py:75-88)
            self.page_bytes += 2 * (tokens_per_block // stats_sub_chunk) * num_kv_heads * head_dim * 2
        self.pool = np.full((num_pages, self.page_bytes), fill, np.uint8)

    def data(self, page):
        return self.pool[page, : self.bytes_per_seq].reshape(self.H, self.TPB, self.D // 2)

    def scales(self, page):
        o = self.bytes_per_seq
        return self.pool[page, o: o + self.H * self.TPB * 2].view(F16).reshape(self.H, self.TPB)

    def zeros(self, page):
        o = self.bytes_per_seq + self.H * self.TPB * 2
        return self.pool[page, o: o + self.H * self.TPB * 2].view(F16).reshape(self.H, self.TPB)

    def write_token(self, page, slot, head, x_h, decode=False):
        sc, ze = kv4_quant_params(x_h)
        self.data(page)[head, slot] = kv4_quantize(x_h, sc, ze)
        self.scales(page)[head, slot] = sc
        self.zeros(page)[head, slot] = ze

    def read_row(self, page, head, slot):
        return kv4_dequant(self.data(page)[head, slot], self.scales(page)[head, slot], self.zeros(page)[head, slot])

    def read_tokens(self, table_row, head, n, fp16_math=True):
        """Dequantized [n, D] float32 for logical tokens 0..n-1 of one sequence (vectorised gather)."""
        if n <= 0:
            return np.empty((0, self.D), F32)
        t = np.arange(n, dtype=np.int64)
        pg = np.asarray(table_row, np.int64)[t // self.TPB]
        sl = t % self.TPB
        flat = self.pool.reshape(-1)
        row = pg * self.pool.shape[1]
        d_off = row + (head * self.TPB + sl) * (self.D // 2)
        packed = flat[d_off[:, None] + np.arange(self.D // 2, dtype=np.int64)[None, :]]
        s_off = row + self.bytes_per_seq + (head * self.TPB + sl) * 2
        z_off = s_off + self.H * self.TPB * 2
        two = np.arange(2, dtype=np.int64)[None, :]
        sc = np.ascontiguousarray(flat[s_off[:, None] + two]).view(F16)[:, 0]
        ze = np.ascontiguousarray(flat[z_off[:, None] + two]).view(F16)[:, 0]
        return kv4_dequant(packed, sc, ze, fp16_math)


def prefill_write(qkv_h, seq_lens, k_cache: PagedKV4, v_cache: PagedKV4, k_table, v_table,
                  num_heads, num_kv_heads, head_dim, rope_base, rope_scale_factor=1.0):
    """apply_bias_rope_update_kv_cache (dense retrieval heads, no bias, neox RoPE,
    KV4 + zeros): RoPE q and k IN PLACE in qkv (STORE_QKV=true), quantize the
    post-RoPE k and the raw v per (token, kv head), write pages + tails.
    qkv fp16 [tokens, (Hq+2Hkv)*D] unpadded, sequences back to back.
    rotary scale type is LINEAR: angle uses pos/rope_scale_factor
    (fine_grained_common/update_kv_cache.cu:75, applyBiasRopeUpdateKVCache.h:596).
    Returns the updated qkv."""
    qkv = np.array(qkv_h, dtype=F16, copy=True)
    D, Hq, Hk = head_dim, num_heads, num_kv_heads
    tok = 0
    for b, L in enumerate(seq_lens):
        for p in range(int(L)):
            row = qkv[tok]
            q = row[: Hq * D].reshape(Hq, D)
            k = row[Hq * D: (Hq + Hk) * D].reshape(Hk, D)
            v = row[(Hq + Hk) * D:].reshape(Hk, D)
            q[:] = rope_neox(q, np.full((Hq,), p), rope_base, 1.0 / rope_scale_factor)
            k[:] = rope_neox(k, np.full((Hk,), p), rope_base, 1.0 / rope_scale_factor)
            for h in range(Hk):
                k_cache.write_token(int(k_table[b][p // k_cache.TPB]), p % k_cache.TPB, h, k[h])
                v_cache.write_token(int(v_table[b][p // v_cache.TPB]), p % v_cache.TPB, h, v[h])
            tok += 1
    return qkv


def decode_attention(q_h, k_h, v_h, lengths, k_cache: PagedKV4, v_cache: PagedKV4, k_table, v_table,
                     rope_base, emulate_fp16: bool = False):
    """single_query_attention (pure dense, KV4 + zeros, neox RoPE, Dh = rotary dim).

    q fp16 [B,Hq,D], k,v fp16 [B,Hkv,D]; lengths[b] = context length INCLUDING
    the current token; tlen = lengths[b]-1 is both the RoPE position and the
    append slot.  Side effect: quantized post-RoPE k and raw v of the current
    token are appended to the caches (scale/zero tails included).
    History keys/values come from the quantized cache (fp16 dequant), the
    current token enters un-quantized.  softmax in f32 with 1/(sum+1e-6);
    emulate_fp16=True additionally rounds probabilities to fp16 before P.V and
    does the history q.k products in fp16 pairs as the reference does
    (Template.hpp:451-467,1794-1845); emulate_fp16=False is the f32 reference
    the HIP kernel is compared against (rtol 1e-3).
    Returns out fp16 [B,Hq,D].
    """
    q_h = np.asarray(q_h, F16)
    B, Hq, D = q_h.shape
    Hk = k_h.shape[1]
    g = Hq // Hk
    inv_sqrt = F32(1.0 / np.sqrt(F32(D)))
    out = np.zeros((B, Hq, D), F16)
    for b in range(B):
        tlen = int(lengths[b]) - 1
        qr = rope_neox(q_h[b], np.full((Hq,), tlen), rope_base)          # fp16
        kr = rope_neox(k_h[b], np.full((Hk,), tlen), rope_base)          # fp16
        vr = np.asarray(v_h[b], F16)
        for hk in range(Hk):
            Kc = k_cache.read_tokens(k_table[b], hk, tlen)               # history, before append
            Vc = v_cache.read_tokens(v_table[b], hk, tlen)
            for hq in range(hk * g, (hk + 1) * g):
                qf = qr[hq].astype(F32)
                if emulate_fp16 and tlen > 0:
                    prod = (qr[hq][None, :].astype(F32) * Kc).astype(F16).astype(F32)
                    s_hist = prod.sum(axis=1).astype(F32)
                else:
                    s_hist = (Kc @ qf).astype(F32) if tlen > 0 else np.zeros((0,), F32)
                s_cur = F32(np.dot(qf.astype(np.float64), kr[hk].astype(np.float64)))
                s = (np.concatenate([s_hist, [s_cur]]).astype(F32) * inv_sqrt).astype(F32)
                m = s.max()
                e = np.exp((s - m).astype(F32)).astype(F32)
                p = (e * (F32(1.0) / (e.sum(dtype=F32) + F32(1e-6)))).astype(F32)
                if emulate_fp16:
                    p = p.astype(F16).astype(F32)
                vals = np.concatenate([Vc, vr[hk][None, :].astype(F32)], axis=0)
                o = (p[:, None].astype(np.float64) * vals.astype(np.float64)).sum(axis=0)
                out[b, hq] = o.astype(F32).astype(F16)
            # append current token (quantized) -- after the reads above
            k_cache.write_token(int(k_table[b][tlen // k_cache.TPB]), tlen % k_cache.TPB, hk, kr[hk])
            v_cache.write_token(int(v_table[b][tlen // v_cache.TPB]), tlen % v_cache.TPB, hk, vr[hk])
    return out


# ---- LServe dynamic sparsity (sparse_utils/ContextPool, sparse_utils/KVPageSelector) --------------------
def stats_page_bytes(num_heads_in_pool, head_dim, tokens_per_block, sub_chunk, row_bytes=None):
    """K page with min/max statistics (cache_engine.py:75-88).  row_bytes = bytes of one token row of one head
    (Dh/2 for KV4, the default; Dh for KV8): the statistics sit behind data + the 4 B/token-head tail."""
    rb = head_dim // 2 if row_bytes is None else row_bytes
    return num_heads_in_pool * tokens_per_block * (rb + 4) + \
        2 * (tokens_per_block // sub_chunk) * num_heads_in_pool * head_dim * 2


def pool_views(pool_row, num_heads_in_pool, head_dim, tokens_per_block, sub_chunk, row_bytes=None):
    """(kmax, kmin) fp16 views [subs][H][D] of one K page (uint8 row)."""
    rb = head_dim // 2 if row_bytes is None else row_bytes
    base = num_heads_in_pool * tokens_per_block * (rb + 4)
    subs = tokens_per_block // sub_chunk
    n = subs * num_heads_in_pool * head_dim
    kmax = pool_row[base: base + 2 * n].view(F16).reshape(subs, num_heads_in_pool, head_dim)
    kmin = pool_row[base + 2 * n: base + 4 * n].view(F16).reshape(subs, num_heads_in_pool, head_dim)
    return kmax, kmin


def paged_min_max_pool(k_h, cu_seqlens, pooling_heads_idx, pool, k_table, tokens_per_block, sub_chunk,
                       row_bytes=None):
    """context_pool_kernel.cu:33-71: per (sequence, pooled head, sub-chunk) elementwise max/min of the keys
    (tokens past the end repeat the last one; a sub-chunk is stored only if its first token exists).
    pool: uint8 [pages, stats_page_bytes]; k_table [B][pages] page indices."""
    k_h = np.asarray(k_h, F16)
    H = len(pooling_heads_idx)
    D = k_h.shape[2]
    for b in range(len(cu_seqlens) - 1):
        s0, s1 = int(cu_seqlens[b]), int(cu_seqlens[b + 1])
        L = s1 - s0
        for r, hin in enumerate(pooling_heads_idx):
            for c in range((L + sub_chunk - 1) // sub_chunk):
                t0 = c * sub_chunk
                toks = np.minimum(np.arange(t0, t0 + sub_chunk), L - 1)
                blk = k_h[s0 + toks, int(hin)]
                page = int(k_table[b][t0 // tokens_per_block])
                kmax, kmin = pool_views(pool[page], H, D, tokens_per_block, sub_chunk, row_bytes)
                sc = (t0 % tokens_per_block) // sub_chunk
                kmax[sc, r] = blk.max(axis=0)
                kmin[sc, r] = blk.min(axis=0)


def page_selector(q_h, lengths, retrieval_head_flags, head_rank_table, pool, k_table, num_kv_heads,
                  num_retrieval_kv_heads, tokens_per_block, sub_chunk, rope_base, rope_scale=1.0, row_bytes=None):
    """KVPageSelectorTemplate.hpp:482-493: score[b,h,c] = sum_d max(h(q_d*kmax_d), h(q_d*kmin_d)) with fp16
    products; q rotated at position lengths[b]-1.  Streaming heads stay zero.  Returns fp16 [B,Hq,padded]."""
    q_h = np.asarray(q_h, F16)
    B, Hq, D = q_h.shape
    g = Hq // num_kv_heads
    tmax = int(max(lengths)) - 1
    n_sub_max = (tmax + sub_chunk - 1) // sub_chunk
    grp = tokens_per_block // sub_chunk
    padded = (n_sub_max + grp - 1) // grp * grp
    out = np.zeros((B, Hq, padded), F16)
    for b in range(B):
        tlen = int(lengths[b]) - 1
        qr = rope_neox(q_h[b], np.full((Hq,), tlen), rope_base, rope_scale)
        for h in range(Hq):
            hk = h // g
            if int(retrieval_head_flags[hk]) == 0:
                continue
            rank = int(head_rank_table[hk])
            for c in range((tlen + sub_chunk - 1) // sub_chunk):
                page = int(k_table[b][(c * sub_chunk) // tokens_per_block])
                kmax, kmin = pool_views(pool[page], num_retrieval_kv_heads, D, tokens_per_block, sub_chunk, row_bytes)
                a = (qr[h] * kmax[c % grp, rank]).astype(F16)
                bb = (qr[h] * kmin[c % grp, rank]).astype(F16)
                out[b, h, c] = np.maximum(a, bb).astype(np.float64).sum()
    return out


# ---- LServe fine-grained caches: retrieval + streaming heads ------------------------------------------
# Citations: fused_attention_fine_grained/fine_grained_common/applyBiasRopeUpdateKVCache.h:296-311
# (which tokens a head stores), common/kvCacheUtils.h:119-126 (ring of streaming pages),
# fused_attention_fine_grained/dense_attention/decoderMaskedMultiheadAttentionTemplate.hpp:1475,1486-1490,1537
# (which cached tokens a streaming head attends to), sparse_attention/...Template.hpp:1566-1570,1632-1641
# (dynamic page indirection), :1414-1429 (statistics update on append), SURVEY.md Appendix B.

def ring_block(blk: int, sink_blocks: int, local_blocks: int) -> int:
    return blk if blk < sink_blocks else sink_blocks + (blk - sink_blocks) % local_blocks


class FineGrainedKV:
    """The two page pools of one layer (PagedKV4 pools, or oracle.kv8.PagedKV8 pools for the per_tensor KV8
    family -- same head classes, rings and page selection): retrieval (K pages optionally with min/max statistics) and streaming
    (ring of sink_blocks + local_blocks pages per sequence).  Tables hold page indices [B][blocks]."""

    def __init__(self, retr_k, retr_v, retr_k_table, retr_v_table, strm_k, strm_v, strm_k_table, strm_v_table,
                 flags, rank, sink, local, sink_blocks, local_blocks, sub_chunk=0):
        self.retr_k, self.retr_v, self.retr_k_table, self.retr_v_table = retr_k, retr_v, retr_k_table, retr_v_table
        self.strm_k, self.strm_v, self.strm_k_table, self.strm_v_table = strm_k, strm_v, strm_k_table, strm_v_table
        self.flags, self.rank = [int(f) for f in flags], [int(r) for r in rank]
        self.sink, self.local, self.sink_blocks, self.local_blocks = sink, local, sink_blocks, local_blocks
        self.sub_chunk = sub_chunk

    def locate(self, b, hk, t):
        """(k cache, v cache, k page, v page, slot) of logical token t of kv head hk."""
        if self.flags[hk]:
            blk = t // self.retr_k.TPB
            return (self.retr_k, self.retr_v, int(self.retr_k_table[b][blk]), int(self.retr_v_table[b][blk]),
                    t % self.retr_k.TPB)
        blk = ring_block(t // self.strm_k.TPB, self.sink_blocks, self.local_blocks)
        return (self.strm_k, self.strm_v, int(self.strm_k_table[b][blk]), int(self.strm_v_table[b][blk]),
                t % self.strm_k.TPB)

    def write(self, b, hk, t, k_row, v_row, update_stats=False, decode=False):
        kc, vc, kp, vp, slot = self.locate(b, hk, t)
        kc.write_token(kp, slot, self.rank[hk], k_row, decode=decode)
        vc.write_token(vp, slot, self.rank[hk], v_row, decode=decode)
        if update_stats and self.flags[hk] and self.sub_chunk:
            kmax, kmin = pool_views(kc.pool[kp], kc.H, kc.D, kc.TPB, self.sub_chunk, kc.row_bytes)
            sc = slot // self.sub_chunk
            kf = np.asarray(k_row, F16)
            kmax[sc, self.rank[hk]] = np.fmax(kmax[sc, self.rank[hk]], kf)
            kmin[sc, self.rank[hk]] = np.fmin(kmin[sc, self.rank[hk]], kf)

    def read(self, b, hk, toks):
        """Dequantised K, V float32 [len(toks), D] of the logical tokens `toks`."""
        D = self.retr_k.D if self.flags[hk] else self.strm_k.D
        K = np.zeros((len(toks), D), F32)
        V = np.zeros((len(toks), D), F32)
        r = self.rank[hk]
        for i, t in enumerate(toks):
            kc, vc, kp, vp, slot = self.locate(b, hk, int(t))
            K[i] = kc.read_row(kp, r, slot)
            V[i] = vc.read_row(vp, r, slot)
        return K, V


def prefill_write_fine_grained(qkv_h, seq_lens, fg: FineGrainedKV, num_heads, num_kv_heads, head_dim, rope_base,
                               rope_scale_factor=1.0):
    """apply_bias_rope_update_kv_cache with both head classes: RoPE q,k in place; a retrieval head stores every
    token, a streaming head only pos < sink or pos >= len - local (in its ring)."""
    qkv = np.array(qkv_h, dtype=F16, copy=True)
    D, Hq, Hk = head_dim, num_heads, num_kv_heads
    tok = 0
    for b, L in enumerate(seq_lens):
        L = int(L)
        for p in range(L):
            row = qkv[tok]
            q = row[: Hq * D].reshape(Hq, D)
            k = row[Hq * D: (Hq + Hk) * D].reshape(Hk, D)
            v = row[(Hq + Hk) * D:].reshape(Hk, D)
            q[:] = rope_neox(q, np.full((Hq,), p), rope_base, 1.0 / rope_scale_factor)
            k[:] = rope_neox(k, np.full((Hk,), p), rope_base, 1.0 / rope_scale_factor)
            for h in range(Hk):
                if fg.flags[h] or p < fg.sink or p >= L - fg.local:
                    fg.write(b, h, p, k[h], v[h])
            tok += 1
    return qkv


def attended_tokens(fg: FineGrainedKV, b, hk, hq, tlen, dyn_pages=None):
    """Logical indices of the cached tokens q head hq attends, in kernel order."""
    if fg.flags[hk]:
        if dyn_pages is None:
            return np.arange(tlen, dtype=np.int64)
        tpb = fg.retr_k.TPB
        P = dyn_pages.shape[-1]
        # sparse_attention/decoderMaskedMultiheadAttentionTemplate.hpp:1568 counts (tlen-1) % tpb + 1 tokens for the
        # last selected page; the reference's Python (decoding_attention.py:132-142) puts the page of the CURRENT
        # token there, which holds tlen - page*tpb cached tokens: the same number except when tlen % tpb == 0, where
        # the page is still empty and upstream's softmax reads score slots it never wrote (:1737-1744 vs :1962,
        # undefined).  The restatement uses the number of tokens the page really holds.
        pages = np.asarray(dyn_pages[b, hq], np.int64)
        in_last = int(min(tpb, max(0, tlen - int(pages[P - 1]) * tpb)))
        nvirt = (P - 1) * tpb + in_last
        i = np.arange(nvirt, dtype=np.int64)
        return pages[i // tpb] * tpb + i % tpb
    valid = min(fg.sink + fg.local - 1, tlen)
    gap = tlen - valid
    i = np.arange(valid, dtype=np.int64)
    return np.where(i < fg.sink, i, i + gap)


def decode_attention_fine_grained(q_h, k_h, v_h, lengths, fg: FineGrainedKV, rope_base, rope_scale_factor=1.0,
                                  dyn_pages=None):
    """fused_attention_fine_grained_{dense,sparse}.single_query_attention (KV4 + zeros).  The current token
    enters un-quantised and is appended (quantised) to the pool of its head class; with dyn_pages (sparse
    variant) the appended key is folded into the page statistics.  f64 softmax reference -> fp16 [B,Hq,D]."""
    q_h = np.asarray(q_h, F16)
    B, Hq, D = q_h.shape
    Hk = k_h.shape[1]
    g = Hq // Hk
    inv_sqrt = 1.0 / np.sqrt(D)
    out = np.zeros((B, Hq, D), F16)
    for b in range(B):
        tlen = int(lengths[b]) - 1
        qr = rope_neox(q_h[b], np.full((Hq,), tlen), rope_base, 1.0 / rope_scale_factor)
        kr = rope_neox(k_h[b], np.full((Hk,), tlen), rope_base, 1.0 / rope_scale_factor)
        vr = np.asarray(v_h[b], F16)
        for hk in range(Hk):
            cache = {}
            for hq in range(hk * g, (hk + 1) * g):
                toks = attended_tokens(fg, b, hk, hq, tlen, dyn_pages)
                key = toks.tobytes()
                if key not in cache:
                    cache[key] = fg.read(b, hk, toks)
                Kc, Vc = cache[key]
                qf = qr[hq].astype(np.float64)
                s = np.concatenate([Kc.astype(np.float64) @ qf, [np.dot(qf, kr[hk].astype(np.float64))]]) * inv_sqrt
                e = np.exp(s - s.max())
                p = e / (e.sum() + 1e-6)
                vals = np.concatenate([Vc.astype(np.float64), vr[hk][None, :].astype(np.float64)], axis=0)
                out[b, hq] = (p[:, None] * vals).sum(axis=0).astype(F16)
            fg.write(b, hk, tlen, kr[hk], vr[hk], update_stats=dyn_pages is not None, decode=True)
    return out


# ---- host-side logic of the LServe path (pure Python / torch upstream), pinned by tests/golden/host_logic.json -------
def select_topk_pages(stats, tokens_per_block, sub_chunk, budget_tokens, timestep):
    """decoding_attention.py:88-142 (DecodingAttentionWrapper.dynamic_select_topk_pages) after the selector kernel:
    stats [B, Hq, padded_sub_chunks] -> int32 [B, Hq, P].  timestep = history length (current token excluded).
    timestep <= budget: every page 0 .. timestep // tpb.  Otherwise the per-page score is the maximum over the page's
    sub-chunks, the k = min(max(3, budget // tpb), total_pages) - 1 best pages among all but the newest are taken in
    descending score order (ties: torch.topk leaves the order unspecified; here: lower page first) and the newest page
    (total_pages - 1, the page of the current token) is appended."""
    stats = np.asarray(stats)
    B, Hq = stats.shape[:2]
    if timestep <= budget_tokens:
        pages = np.arange(0, timestep // tokens_per_block + 1, dtype=np.int32)
        return np.broadcast_to(pages, (B, Hq, pages.size)).copy()
    subs = tokens_per_block // sub_chunk
    page_scores = stats.reshape(B, Hq, -1, subs).max(axis=-1)
    total = page_scores.shape[-1]
    k = min(max(3, min(budget_tokens, timestep) // tokens_per_block), total) - 1
    order = np.argsort(-page_scores[:, :, : total - 1].astype(np.float32), axis=-1, kind="stable")[:, :, :k]
    newest = np.full((B, Hq, 1), total - 1, np.int64)
    return np.concatenate([order, newest], axis=-1).astype(np.int32)


def head_classes(full_attention_heads, num_heads):
    """ctx_attn_init.py:28-81 for one layer: kv-head flags (1 = retrieval / dense head, 0 = streaming head) ->
    dict(head_mask_type int32 [Hq] (0 dense, -1 streaming; None when every head is dense), retrieval_head_flags,
    head_rank_table (index of a head inside its class), pooling_heads_idx (the retrieval heads))."""
    flags = np.asarray(full_attention_heads, np.int32)
    rep = num_heads // flags.size
    hm = None if (flags == 1).all() else np.where(np.repeat(flags, rep) == 0, -1, 0).astype(np.int32)
    rank = np.empty_like(flags)
    rank[flags == 0] = np.arange((flags == 0).sum(), dtype=np.int32)
    rank[flags == 1] = np.arange((flags == 1).sum(), dtype=np.int32)
    return dict(head_mask_type=hm, retrieval_head_flags=flags, head_rank_table=rank,
                pooling_heads_idx=np.nonzero(flags == 1)[0].astype(np.int32))
