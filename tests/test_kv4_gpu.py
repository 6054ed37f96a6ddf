"""Parity of the KV4 cache kernels: prefill writer (bit-exact pages + in-place RoPE) and decode
attention (fp16 output within 1e-3 relative of the f32 oracle; appended cache rows bit-exact)."""
import numpy as np
import pytest
import torch

from oracle import kv4
from tests.util import GpuPagedKV, assert_attention_close, assert_f16_equal, dev, to_dev

pytestmark = pytest.mark.gpu

D = 128
ROPE_BASE = 500000.0


def _tables(B, pages_per_seq, seed):
    rng = np.random.default_rng(seed)
    n = B * pages_per_seq
    kidx = rng.permutation(n).reshape(B, pages_per_seq)
    vidx = rng.permutation(n).reshape(B, pages_per_seq)
    return n, kidx, vidx


def _prefill_case(seq_lens, Hq, Hk, seed, scale_factor=1.0):
    import omniserve_backend.fused_attention_fine_grained_dense as fa
    rng = np.random.default_rng(seed)
    B = len(seq_lens)
    max_len = int(max(seq_lens))
    pages = (max_len + 64) // 64 + 1
    n, kidx, vidx = _tables(B, pages, seed)
    kc, vc = kv4.PagedKV4(n, Hk, D, fill=0x5A), kv4.PagedKV4(n, Hk, D, fill=0x5A)
    T = int(sum(seq_lens))
    qkv = rng.standard_normal((T, (Hq + 2 * Hk) * D)).astype(np.float16)
    gk = GpuPagedKV(kc, vc, kidx, vidx)
    want_qkv = kv4.prefill_write(qkv, seq_lens, kc, vc, kidx, vidx, Hq, Hk, D, ROPE_BASE, scale_factor)
    cu = np.concatenate([[0], np.cumsum(seq_lens)]).astype(np.int32)
    pad = fa.compute_padding_offsets(to_dev(cu), max_len, T)
    assert np.array_equal(pad.cpu().numpy(), kv4.compute_padding_offsets(cu, max_len))
    qkv_d = to_dev(qkv)
    lens_d = to_dev(np.asarray(seq_lens, np.int32))
    flags = to_dev(np.ones(Hk, np.int32)); rank = to_dev(np.arange(Hk, dtype=np.int32))
    fa.apply_bias_rope_update_kv_cache(
        qkv_d, lens_d, None, pad, gk.table, None, flags, rank, Hq, Hk, max_len, 64, Hk * D // 2, 0,
        0, 0, 0, 0, Hk, 0, D, ROPE_BASE, scale_factor, 1 << 20, True, True, True)
    torch.cuda.synchronize()
    assert_f16_equal(qkv_d, want_qkv, "qkv after in-place RoPE")
    kp, vp = gk.pools()
    assert np.array_equal(kp, kc.pool), "K pages differ"
    assert np.array_equal(vp, vc.pool), "V pages differ"
    return kc, vc, kidx, vidx, gk


@pytest.mark.parametrize("seq_lens,Hq,Hk", [([5], 4, 1), ([64, 1, 130], 8, 2), ([200, 77], 32, 8), ([1], 8, 8)])
def test_prefill_write(seq_lens, Hq, Hk):
    _prefill_case(seq_lens, Hq, Hk, seed=len(seq_lens) + Hq)


def test_prefill_write_linear_rope_scaling():
    _prefill_case([70, 3], 8, 2, seed=11, scale_factor=4.0)


def _decode_case(hist_lens, Hq, Hk, seed, steps=1):
    import omniserve_backend.fused_attention_pure_dense as fa
    rng = np.random.default_rng(seed)
    B = len(hist_lens)
    kc, vc, kidx, vidx, gk = _prefill_case([max(int(h), 1) for h in hist_lens], Hq, Hk, seed)
    # sequences whose wanted history is 0 still got one prefill token: treat lengths explicitly
    lens = np.asarray([max(int(h), 1) for h in hist_lens], np.int32)
    for step in range(steps):
        lens = lens + 1                                   # context length incl. the current token
        qkv = rng.standard_normal((B, (Hq + 2 * Hk) * D)).astype(np.float16)
        q = qkv[:, : Hq * D].reshape(B, Hq, D)
        k = qkv[:, Hq * D:(Hq + Hk) * D].reshape(B, Hk, D)
        v = qkv[:, (Hq + Hk) * D:].reshape(B, Hk, D)
        want = kv4.decode_attention(q, k, v, lens, kc, vc, kidx, vidx, ROPE_BASE)
        qkv_d = to_dev(qkv)                               # strided views of the fused buffer
        qd = qkv_d[:, : Hq * D].view(B, Hq, D)
        kd = qkv_d[:, Hq * D:(Hq + Hk) * D].view(B, Hk, D)
        vd = qkv_d[:, (Hq + Hk) * D:].view(B, Hk, D)
        out = fa.single_query_attention(qd, kd, vd, gk.table, to_dev(lens), None, 65536, 64, Hk * D // 2,
                                        int(lens.max()), D, ROPE_BASE, True, True, True)
        torch.cuda.synchronize()
        got = out.cpu().numpy().astype(np.float32)
        ref = want.astype(np.float32)
        assert_attention_close(got, ref, "decode attention, history %s, step %d" % (list(hist_lens), step))
        kp, vp = gk.pools()
        assert np.array_equal(kp, kc.pool), "K pages differ after append (step %d)" % step
        assert np.array_equal(vp, vc.pool), "V pages differ after append (step %d)" % step


@pytest.mark.parametrize("hist_lens,Hq,Hk", [([5], 4, 1), ([63, 64, 65], 8, 2), ([200, 17, 130, 1], 32, 8),
                                              ([300], 8, 8), ([90, 33], 16, 2)])
def test_decode_attention(hist_lens, Hq, Hk):
    _decode_case(hist_lens, Hq, Hk, seed=sum(hist_lens) + Hq)


@pytest.mark.parametrize("hist_lens,Hq,Hk,steps", [([70, 5, 129], 16, 2, 2), ([300, 64], 64, 8, 1), ([1400, 1030, 1], 8, 1, 1),
                                                    ([2100], 16, 2, 1), ([33], 24, 3, 1)])
def test_decode_attention_eight_q_heads_per_kv_head(hist_lens, Hq, Hk, steps):
    """Head groups of 8 (Llama-2/3-70B: 64 / 8, one TP = 8 rank: 8 / 1) run as ONE workgroup per (split, kv head, sequence) with
    eight head columns (kv4_decode_flash_kernel<8, ...>): short and long sweeps (both loop forms), one split and several (the
    merge launch), the append crossing a page; 24 / 3 stays on the four-column form."""
    _decode_case(hist_lens, Hq, Hk, seed=sum(hist_lens) + Hq, steps=steps)


def test_decode_attention_multi_step_crossing_page():
    _decode_case([62, 127], 8, 2, seed=3, steps=4)


def test_decode_attention_long_context_many_splits():
    _decode_case([1500, 1030], 32, 8, seed=4)


def test_decode_attention_very_long_context():
    """> 64 x 2048 tokens: the KV split count grows past 64 (LServe contexts).  Random page bytes with sane
    scale / zero tails stand in for a prefill (the oracle reads the same pages)."""
    import omniserve_backend.fused_attention_pure_dense as fa
    rng = np.random.default_rng(77)
    Hq, Hk, T = 4, 1, 140000
    pages = T // 64 + 2
    kc, vc = kv4.PagedKV4(pages, Hk, D), kv4.PagedKV4(pages, Hk, D)
    for c in (kc, vc):
        c.pool[:] = rng.integers(0, 256, c.pool.shape, dtype=np.uint8)
        for p in range(pages):
            c.scales(p)[:] = (0.02 + 0.05 * rng.random((Hk, 64))).astype(np.float16)
            c.zeros(p)[:] = (6.0 + 3.0 * rng.random((Hk, 64))).astype(np.float16)
    kidx = rng.permutation(pages).reshape(1, pages)
    vidx = rng.permutation(pages).reshape(1, pages)
    gk = GpuPagedKV(kc, vc, kidx, vidx)
    lens = np.asarray([T + 1], np.int32)
    q = (0.3 * rng.standard_normal((1, Hq, D))).astype(np.float16)
    k = rng.standard_normal((1, Hk, D)).astype(np.float16)
    v = rng.standard_normal((1, Hk, D)).astype(np.float16)
    want = kv4.decode_attention(q, k, v, lens, kc, vc, kidx, vidx, ROPE_BASE)
    out = fa.single_query_attention(to_dev(q), to_dev(k), to_dev(v), gk.table, to_dev(lens), None, 1 << 20, 64,
                                    Hk * D // 2, T + 1, D, ROPE_BASE, True, True, True)
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float32)
    ref = want.astype(np.float32)
    assert_attention_close(got, ref, "decode attention, T = %d" % T)
    kp, vp = gk.pools()
    assert np.array_equal(kp, kc.pool) and np.array_equal(vp, vc.pool)
