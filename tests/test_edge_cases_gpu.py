"""Edge cases of the reference's arithmetic that SURVEY.md Appendix A lists (A.2 all-zero row in the per-token
quantisers, A.3 the KV nibble is masked `& 0xF` and NOT clamped, A.7 constant K/V rows -> zero range -> inf / NaN
parameters), plus saturating inputs (+-65504, fp16-overflowing row sums) and the exact BASELINE shapes of the KV4
decode attention (B = 16 / 64, 32 q heads over 8 kv heads, T = 1024 / 1535).  HIP kernels vs oracle/ on the same inputs.

NaN payloads: the reference's CUDA build, numpy on x86 and the gfx950 VALU each produce their own quiet-NaN bit pattern
for 0/0 and inf-inf; fp16 words that are NaN on both sides are compared as "NaN", everything else bit for bit."""
import numpy as np
import pytest
import torch

from oracle import elementwise as oe
from oracle import kv4
from tests.util import GpuPagedKV, assert_f16_equal, dev, to_dev

pytestmark = pytest.mark.gpu

D = 128
ROPE_BASE = 500000.0


def _f16_equal_nan_aware(got, want, what):
    g = np.ascontiguousarray(got.detach().cpu().numpy() if torch.is_tensor(got) else got).view(np.uint16).copy()
    w = np.ascontiguousarray(want).view(np.uint16).copy()
    gn = (g & 0x7FFF) > 0x7C00
    wn = (w & 0x7FFF) > 0x7C00
    assert np.array_equal(gn, wn), "%s: NaN positions differ" % what
    g[gn] = 0x7E00
    w[wn] = 0x7E00
    bad = g != w
    assert not bad.any(), "%s: %d fp16 words differ, first at %s" % (what, bad.sum(), np.argwhere(bad)[:4].tolist())


# ---------------------------------------------------------------------------------------------------------------
# A.2 / saturation: per-token quantisers
# ---------------------------------------------------------------------------------------------------------------
def _edge_rows(hidden, seed):
    """rows: 0 all zero | 1 single non-zero element | 2 all +65504 | 3 alternating +-65504 | 4 one 65504 among noise |
    5 constant small | 6 a row whose f32 sum overflows fp16 (sum -> +inf) | 7 negative zeros | 8 ordinary"""
    rng = np.random.default_rng(seed)
    x = np.zeros((9, hidden), np.float16)
    x[1, hidden // 3] = np.float16(-0.37)
    x[2] = np.float16(65504.0)
    x[3, 0::2] = np.float16(65504.0)
    x[3, 1::2] = np.float16(-65504.0)
    x[4] = rng.standard_normal(hidden).astype(np.float16)
    x[4, 5] = np.float16(65504.0)
    x[5] = np.float16(0.011)
    x[6] = np.float16(700.0) + rng.standard_normal(hidden).astype(np.float16)
    x[7] = np.float16(-0.0)
    x[8] = (rng.standard_normal(hidden) * 3).astype(np.float16)
    return x


@pytest.mark.parametrize("hidden", [4096, 14336, 128, 28672])
@pytest.mark.parametrize("fuse", [False, True])
def test_quant_edge_rows(hidden, fuse):
    """invoke_quant[_fuse_sum]: no amax floor (fused_kernels.cu:108-131) -> the all-zero row divides by zero
    (scale 0, 127/0 = inf, 0*inf = NaN -> code 0); saturating rows; row sums that overflow fp16."""
    import omniserve_backend.fused_kernels as fk
    x = _edge_rows(hidden, hidden)
    T = x.shape[0]
    out = torch.full((T, hidden), 77, dtype=torch.int8, device=dev())
    scale = torch.empty((T,), dtype=torch.float16, device=dev())
    ssum = torch.empty((T,), dtype=torch.float16, device=dev())
    if fuse:
        fk.invoke_quant_fuse_sum(out, to_dev(x), ssum, scale)
    else:
        fk.invoke_quant(out, to_dev(x), scale)
    torch.cuda.synchronize()
    with np.errstate(all="ignore"):
        q, s, sm = oe.quant_per_token(x, fuse)
    assert np.array_equal(out.cpu().numpy(), q)
    assert (q[0] == 0).all() and (q[7] == 0).all() and float(s[0]) == 0.0
    _f16_equal_nan_aware(scale, s, "scale")
    if fuse:
        _f16_equal_nan_aware(ssum, sm, "sum")
        assert np.isinf(sm[2].astype(np.float32)) and (hidden < 256 or np.isinf(sm[6].astype(np.float32)))


@pytest.mark.parametrize("hidden", [4096, 8192, 128])
@pytest.mark.parametrize("fuse", [False, True])
def test_rms_norm_general_edge_rows(hidden, fuse):
    """generalLayerNorm: all-zero and constant rows (x - mean == 0 everywhere: amax falls to its 1e-6 floor,
    layernorm_kernels.cu:279-291), +-65504 (x^2 sums near 2e13 in f32)."""
    import omniserve_backend.layernorm_ops as ln
    x = _edge_rows(hidden, hidden + 1)
    g = (1.0 + 0.1 * np.random.default_rng(1).standard_normal(hidden)).astype(np.float16)
    T = x.shape[0]
    out = torch.full((T, hidden), 77, dtype=torch.int8, device=dev())
    scale = torch.empty((T,), dtype=torch.float16, device=dev())
    ssum = torch.empty((T,), dtype=torch.float16, device=dev())
    if fuse:
        ln.rms_norm_general_fuse_sum(out, to_dev(x), to_dev(g), ssum, scale, 1e-5, True)
    else:
        ln.rms_norm_general(out, to_dev(x), to_dev(g), scale, 1e-5, True)
    torch.cuda.synchronize()
    with np.errstate(all="ignore"):
        q, s, sm = oe.rms_norm_general(x, g, 1e-5, fuse)
    assert np.array_equal(out.cpu().numpy(), q)
    _f16_equal_nan_aware(scale, s, "scale")
    if fuse:
        _f16_equal_nan_aware(ssum, sm, "sum")


@pytest.mark.parametrize("d", [14336, 128, 28672])
@pytest.mark.parametrize("with_sum", [True, False])
def test_fused_silu_quant_edge_rows(d, with_sum):
    """The fused SiLU*mul quantiser (fused extension) on the same edge rows as the two-kernel sequence it replaces:
    gate = 0 / up = anything (all-zero product row), saturating products (65504 * 65504 -> inf in fp16 -> amax inf)."""
    import omniserve_backend.activation_ops as act
    import omniserve_backend.fused_kernels as fk
    from omniserve_amd.backend import fused_ext
    gate, up = _edge_rows(d, 2 * d), _edge_rows(d, 2 * d + 1)
    up[0] = np.float16(3.0)          # silu(0) * 3 = 0: all-zero row through the activation
    x = to_dev(np.concatenate([gate, up], axis=1))
    T = gate.shape[0]
    tmp = torch.empty((T, d), dtype=torch.float16, device=dev())
    act.silu_and_mul(tmp, x)
    q1 = torch.empty((T, d), dtype=torch.int8, device=dev())
    s1 = torch.empty((T,), dtype=torch.float16, device=dev())
    m1 = torch.zeros((T,), dtype=torch.float16, device=dev())
    if with_sum:
        fk.invoke_quant_fuse_sum(q1, tmp, m1, s1)
    else:
        fk.invoke_quant(q1, tmp, s1)
    q2 = torch.empty_like(q1); s2 = torch.empty_like(s1); m2 = torch.zeros_like(s1)
    fused_ext.silu_mul_quant_fuse_sum(q2, x, m2 if with_sum else None, s2)
    torch.cuda.synchronize()
    assert torch.equal(q1, q2)
    _f16_equal_nan_aware(s2, s1.cpu().numpy(), "scale")
    _f16_equal_nan_aware(m2, m1.cpu().numpy(), "sum")
    # and the two-kernel sequence against the oracle on the activation the device produced
    with np.errstate(all="ignore"):
        q, s, sm = oe.quant_per_token(tmp.cpu().numpy(), with_sum)
    assert np.array_equal(q1.cpu().numpy(), q)
    _f16_equal_nan_aware(s1, s, "scale vs oracle")
    assert (q[0] == 0).all()


def test_silu_mul_quant_without_sum_beyond_the_v2_geometry():
    """sum = None with a row longer than the v2 geometry covers (d > 16384: Llama-2-70B's 28672 at TP = 1)
    used to return EINVAL; it must give the codes / scales of the summing form."""
    from omniserve_amd.backend import fused_ext
    rng = np.random.default_rng(5)
    T, d = 3, 20480
    x = to_dev((rng.standard_normal((T, 2 * d)) * 2).astype(np.float16))
    res = []
    for with_sum in (True, False):
        q = torch.empty((T, d), dtype=torch.int8, device=dev())
        sc = torch.empty((T,), dtype=torch.float16, device=dev())
        sm = torch.empty((T,), dtype=torch.float16, device=dev())
        fused_ext.silu_mul_quant_fuse_sum(q, x, sm if with_sum else None, sc)
        torch.cuda.synchronize()
        res.append((q.cpu(), sc.cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1].view(torch.int16), res[1][1].view(torch.int16))


# ---------------------------------------------------------------------------------------------------------------
# A.3 / A.7: KV4 quantiser of the prefill writer and of the decode append
# ---------------------------------------------------------------------------------------------------------------
def _wrap_row():
    """fp16 subnormals k * 2^-24, k = 0..21: range / 15 = 1.4 ulp rounds to a scale of ONE subnormal ulp, so the largest
    element lands on code 21 -> stored nibble 21 & 0xF = 5 (a clamp would store 15): Appendix A.3."""
    r = np.zeros(D, np.float16)
    ks = np.arange(D) % 22
    r[:] = (ks * 2.0 ** -24).astype(np.float16)
    return r


def _special_rows():
    rows = {
        "constant+": np.full(D, 0.75, np.float16),      # max == min > 0: scale 0, zero -inf, codes from NaN -> 0
        "constant-": np.full(D, -1.5, np.float16),
        "zeros": np.zeros(D, np.float16),               # zero = -0/0 = NaN
        "wrap": _wrap_row(),
        "sat": np.where(np.arange(D) % 2 == 0, 65504.0, -65504.0).astype(np.float16),   # range overflows fp16: scale inf
    }
    return rows


def _pages_equal_nan_aware(got_pool, want_pool, cache, what):
    bps = cache.bytes_per_seq
    assert np.array_equal(got_pool[:, :bps], want_pool[:, :bps]), "%s: packed codes differ" % what
    g = np.ascontiguousarray(got_pool[:, bps:]).view(np.float16)
    w = np.ascontiguousarray(want_pool[:, bps:]).view(np.float16)
    _f16_equal_nan_aware(g, w, what + " scale/zero tails")


def test_prefill_writer_special_rows():
    """Single-token sequences (RoPE at position 0 is the identity) whose k and v rows are the special rows."""
    import omniserve_backend.fused_attention_fine_grained_dense as fa
    rows = _special_rows()
    names = list(rows)
    Hq, Hk = 4, 2
    B = len(names)
    seq_lens = [1] * B
    rng = np.random.default_rng(2)
    qkv = rng.standard_normal((B, (Hq + 2 * Hk) * D)).astype(np.float16)
    for b, n in enumerate(names):
        k = qkv[b, Hq * D:(Hq + Hk) * D].reshape(Hk, D)
        v = qkv[b, (Hq + Hk) * D:].reshape(Hk, D)
        k[0] = rows[n]                      # head 0 special, head 1 ordinary
        v[1] = rows[n]
    pages = 2
    n_pages = B * pages
    kidx = rng.permutation(n_pages).reshape(B, pages)
    vidx = rng.permutation(n_pages).reshape(B, pages)
    kc, vc = kv4.PagedKV4(n_pages, Hk, D, fill=0x5A), kv4.PagedKV4(n_pages, Hk, D, fill=0x5A)
    gk = GpuPagedKV(kc, vc, kidx, vidx)
    with np.errstate(all="ignore"):
        want_qkv = kv4.prefill_write(qkv, seq_lens, kc, vc, kidx, vidx, Hq, Hk, D, ROPE_BASE)
    cu = np.arange(B + 1, dtype=np.int32)
    pad = fa.compute_padding_offsets(to_dev(cu), 1, B)
    qkv_d = to_dev(qkv)
    flags = to_dev(np.ones(Hk, np.int32)); rank = to_dev(np.arange(Hk, dtype=np.int32))
    fa.apply_bias_rope_update_kv_cache(qkv_d, to_dev(np.asarray(seq_lens, np.int32)), None, pad, gk.table, None, flags,
                                       rank, Hq, Hk, 1, 64, Hk * D // 2, 0, 0, 0, 0, 0, Hk, 0, D, ROPE_BASE, 1.0,
                                       1 << 20, True, True, True)
    torch.cuda.synchronize()
    assert_f16_equal(qkv_d, want_qkv, "qkv after in-place RoPE")
    kp, vp = gk.pools()
    _pages_equal_nan_aware(kp, kc.pool, kc, "K pages")
    _pages_equal_nan_aware(vp, vc.pool, vc, "V pages")
    # the wrap row really wrapped (code 21 -> nibble 5), i.e. the case is exercised
    b = names.index("wrap")
    stored = vc.data(int(vidx[b][0]))[1, 0]
    codes = np.stack([stored & 0xF, stored >> 4], axis=1).reshape(-1)
    assert codes[21] == 5 and codes[15] == 15 and codes[16] == 0


def test_decode_append_special_rows():
    """The decode kernel's append (RoPE + quantise + page write of the CURRENT token) on the special rows: v rows are
    stored raw, k rows after RoPE at position tlen.  The attention output stays finite (the current token enters
    un-quantised, Appendix A.10) and is compared as usual; pages are compared NaN-aware."""
    import omniserve_backend.fused_attention_pure_dense as fa
    rows = _special_rows()
    names = list(rows)
    Hq, Hk = 8, 2
    B = len(names)
    rng = np.random.default_rng(9)
    hist = [40 + 3 * b for b in range(B)]
    pages = 2
    n_pages = B * pages
    kidx = rng.permutation(n_pages).reshape(B, pages)
    vidx = rng.permutation(n_pages).reshape(B, pages)
    kc, vc = kv4.PagedKV4(n_pages, Hk, D, fill=0x5A), kv4.PagedKV4(n_pages, Hk, D, fill=0x5A)
    T = sum(hist)
    qkv0 = rng.standard_normal((T, (Hq + 2 * Hk) * D)).astype(np.float16)
    kv4.prefill_write(qkv0, hist, kc, vc, kidx, vidx, Hq, Hk, D, ROPE_BASE)
    gk = GpuPagedKV(kc, vc, kidx, vidx)
    lens = np.asarray(hist, np.int32) + 1
    q = rng.standard_normal((B, Hq, D)).astype(np.float16)
    k = rng.standard_normal((B, Hk, D)).astype(np.float16)
    v = rng.standard_normal((B, Hk, D)).astype(np.float16)
    for b, n in enumerate(names):
        v[b, 0] = rows[n]
        if n in ("zeros",):
            k[b, 1] = rows[n]               # RoPE of an all-zero row is all zero
    with np.errstate(all="ignore"):
        want = kv4.decode_attention(q, k, v, lens, kc, vc, kidx, vidx, ROPE_BASE)
    out = fa.single_query_attention(to_dev(q), to_dev(k), to_dev(v), gk.table, to_dev(lens), None, 65536, 64,
                                    Hk * D // 2, int(lens.max()), D, ROPE_BASE, True, True, True)
    torch.cuda.synchronize()
    kp, vp = gk.pools()
    _pages_equal_nan_aware(kp, kc.pool, kc, "K pages after append")
    _pages_equal_nan_aware(vp, vc.pool, vc, "V pages after append")
    got = out.cpu().numpy().astype(np.float32)
    ref = want.astype(np.float32)
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin)
    # (the +-65504 value row weights into the output: compare relative to each head's own largest entry)
    scale = np.abs(np.where(fin, ref, 0)).max(axis=-1, keepdims=True)
    err = np.abs(np.where(fin, got - ref, 0))
    assert (err <= 2e-3 * scale + 1e-6).all(), "append case: max err / head max = %g" % (err / (scale + 1e-30)).max()


# ---------------------------------------------------------------------------------------------------------------
# a9 at the BASELINE shapes: B = 16 and 64, 32 q heads / 8 kv heads, T = 1024 and 1535
# ---------------------------------------------------------------------------------------------------------------
from tests.util import attention_errors  # noqa: E402  (the per-head bar shared by every attention test)


@pytest.mark.parametrize("B,T", [(16, 1024), (16, 1535), (64, 1024), (64, 1535)])
def test_decode_attention_baseline_shapes(B, T):
    """BASELINE configs[1] / configs[2] decode attention: Llama-3-8B heads, every sequence at history length T (first and
    last step of the qserve_benchmark.py protocol).  Pages hold random codes with sane scale / zero tails (a prefill
    through the oracle at 64 x 1535 tokens would take minutes; the oracle reads the same pages).
    Bar: north_star's 1e-3 RELATIVE, taken per (sequence, head) output vector: ||got - ref||_2 <= 1e-3 ||ref||_2 and no
    element further than 1e-3 of that vector's largest entry.  The message carries the measured distances, and the
    distance of the reference's own fp16-emulating arithmetic to the same f32 oracle."""
    import omniserve_backend.fused_attention_pure_dense as fa
    Hq, Hk = 32, 8
    rng = np.random.default_rng(B * 10000 + T)
    pps = (T + 1 + 63) // 64
    n_pages = B * pps
    kc, vc = kv4.PagedKV4(n_pages, Hk, D), kv4.PagedKV4(n_pages, Hk, D)
    for c in (kc, vc):
        c.pool[:] = rng.integers(0, 256, c.pool.shape, dtype=np.uint8)
        for p in range(n_pages):
            c.scales(p)[:] = (0.05 + 0.15 * rng.random((Hk, 64))).astype(np.float16)
            c.zeros(p)[:] = (6.0 + 3.0 * rng.random((Hk, 64))).astype(np.float16)
    kidx = rng.permutation(n_pages).reshape(B, pps)
    vidx = rng.permutation(n_pages).reshape(B, pps)
    gk = GpuPagedKV(kc, vc, kidx, vidx)
    lens = np.full((B,), T + 1, np.int32)
    q = (0.5 * rng.standard_normal((B, Hq, D))).astype(np.float16)
    k = rng.standard_normal((B, Hk, D)).astype(np.float16)
    v = rng.standard_normal((B, Hk, D)).astype(np.float16)
    nb = min(B, 16)                      # the oracle runs on the first 16 sequences (seconds); pages of all are checked
    kc2 = kv4.PagedKV4(1, Hk, D); kc2.pool = kc.pool.copy()
    vc2 = kv4.PagedKV4(1, Hk, D); vc2.pool = vc.pool.copy()
    want = kv4.decode_attention(q[:nb], k[:nb], v[:nb], lens[:nb], kc, vc, kidx, vidx, ROPE_BASE)
    emu = kv4.decode_attention(q[:nb], k[:nb], v[:nb], lens[:nb], kc2, vc2, kidx, vidx, ROPE_BASE, emulate_fp16=True)
    out = fa.single_query_attention(to_dev(q), to_dev(k), to_dev(v), gk.table, to_dev(lens), None, 65536, 64,
                                    Hk * D // 2, T + 1, D, ROPE_BASE, True, True, True)
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float32)
    l2, linf = attention_errors(got[:nb], want)
    l2e, linfe = attention_errors(emu, want)
    msg = ("B=%d T=%d: HIP vs f32 oracle: rel L2 %.3g, max |d| / head max %.3g; fp16-emulating reference arithmetic vs "
           "f32 oracle: rel L2 %.3g, max %.3g" % (B, T, l2, linf, l2e, linfe))
    print(msg)
    assert l2 <= 1e-3 and linf <= 1e-3, msg
    # appended rows of the oracle's sequences: byte-identical
    kp, vp = gk.pools()
    for b in range(nb):
        pk, pv = int(kidx[b][T // 64]), int(vidx[b][T // 64])
        assert np.array_equal(kp[pk], kc.pool[pk]) and np.array_equal(vp[pv], vc.pool[pv]), "appended row, seq %d" % b
    assert np.isfinite(got).all()
