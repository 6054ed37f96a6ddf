"""Parity of quant / norm / activation kernels against the oracle."""
import numpy as np
import pytest
import torch

from oracle import elementwise as oe
from tests.util import assert_f16_equal, dev, f16_ulp_diff, to_dev

pytestmark = pytest.mark.gpu


def _x(tokens, hidden, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((tokens, hidden)) * scale).astype(np.float16)


@pytest.mark.parametrize("tokens,hidden", [(1, 128), (16, 4096), (37, 14336), (5, 1024), (3, 28672), (0, 4096),
                                           (1100, 4096), (1030, 14336)])   # many rows: 128 / 256 threads per row
@pytest.mark.parametrize("fuse", [False, True])
def test_quant(tokens, hidden, fuse):
    import omniserve_backend.fused_kernels as fk
    x = _x(tokens, hidden, tokens + hidden, 3.0)
    out = torch.empty((tokens, hidden), dtype=torch.int8, device=dev())
    scale = torch.empty((tokens,), dtype=torch.float16, device=dev())
    ssum = torch.empty((tokens,), dtype=torch.float16, device=dev())
    if fuse:
        fk.invoke_quant_fuse_sum(out, to_dev(x), ssum, scale)
    else:
        fk.invoke_quant(out, to_dev(x), scale)
    torch.cuda.synchronize()
    if tokens == 0:
        return
    q, s, sm = oe.quant_per_token(x, fuse)
    assert np.array_equal(out.cpu().numpy(), q)
    assert_f16_equal(scale, s, "scale")
    if fuse:
        assert_f16_equal(ssum, sm, "sum")


@pytest.mark.parametrize("tokens,hidden", [(1, 128), (16, 4096), (33, 8192), (7, 5120), (4, 96), (1100, 4096), (1030, 8192)])
@pytest.mark.parametrize("fuse", [False, True])
def test_rms_norm_general(tokens, hidden, fuse):
    import omniserve_backend.layernorm_ops as ln
    x = _x(tokens, hidden, 3 * tokens + hidden, 2.0) + np.float16(0.25)   # non-zero mean matters
    g = (1.0 + 0.1 * np.random.default_rng(1).standard_normal(hidden)).astype(np.float16)
    out = torch.empty((tokens, hidden), dtype=torch.int8, device=dev())
    scale = torch.empty((tokens,), dtype=torch.float16, device=dev())
    ssum = torch.empty((tokens,), dtype=torch.float16, device=dev())
    if fuse:
        ln.rms_norm_general_fuse_sum(out, to_dev(x), to_dev(g), ssum, scale, 1e-5, True)
    else:
        ln.rms_norm_general(out, to_dev(x), to_dev(g), scale, 1e-5, True)
    torch.cuda.synchronize()
    q, s, sm = oe.rms_norm_general(x, g, 1e-5, fuse)
    assert np.array_equal(out.cpu().numpy(), q)
    assert_f16_equal(scale, s, "scale")
    if fuse:
        assert_f16_equal(ssum, sm, "sum")


@pytest.mark.parametrize("tokens,hidden", [(1, 128), (16, 4096), (9, 8192), (1100, 4096)])
def test_rms_norm(tokens, hidden):
    import omniserve_backend.layernorm_ops as ln
    x = _x(tokens, hidden, tokens, 2.0)
    w = (1.0 + 0.1 * np.random.default_rng(2).standard_normal(hidden)).astype(np.float16)
    out = torch.empty((tokens, hidden), dtype=torch.float16, device=dev())
    ln.rms_norm(out, to_dev(x), to_dev(w), 1e-5, False)
    torch.cuda.synchronize()
    assert_f16_equal(out, oe.rms_norm(x, w, 1e-5), "rms_norm")


@pytest.mark.parametrize("tokens,d", [(1, 64), (16, 14336), (5, 100), (33, 4096)])
def test_silu_and_mul(tokens, d):
    import omniserve_backend.activation_ops as act
    x = _x(tokens, 2 * d, tokens + d, 2.0)
    out = torch.empty((tokens, d), dtype=torch.float16, device=dev())
    act.silu_and_mul(out, to_dev(x))
    torch.cuda.synchronize()
    want = oe.silu_and_mul(x)
    # expf on the device vs numpy's differ by an ulp in f32; each of the two fp16 roundings
    # (silu -> fp16, product -> fp16) can then flip by one ulp: <= 2 fp16 ulp, and rarely
    assert f16_ulp_diff(out, want) <= 2
    assert (np.asarray(out.cpu().numpy()).view(np.uint16) != want.view(np.uint16)).mean() < 1e-3


@pytest.mark.parametrize("tokens,hidden", [(16, 4096), (5, 8192), (3, 128), (1100, 4096)])
def test_fused_add_norm_matches_the_two_reference_calls(tokens, hidden):
    import omniserve_backend.layernorm_ops as ln
    from omniserve_amd.backend import fused_ext
    x = _x(tokens, hidden, 11 + tokens, 2.0)
    delta = _x(tokens, hidden, 12 + tokens, 1.0)
    g = (1.0 + 0.1 * np.random.default_rng(1).standard_normal(hidden)).astype(np.float16)
    xs = (x.astype(np.float32) + delta.astype(np.float32)).astype(np.float16)     # torch fp16 add
    q, s, sm = oe.rms_norm_general(xs, g, 1e-5, True)
    res = to_dev(x)
    out = torch.empty((tokens, hidden), dtype=torch.int8, device=dev())
    scale = torch.empty((tokens,), dtype=torch.float16, device=dev())
    ssum = torch.empty((tokens,), dtype=torch.float16, device=dev())
    fused_ext.add_rms_norm_general_fuse_sum(out, res, to_dev(delta), to_dev(g), ssum, scale, 1e-5)
    torch.cuda.synchronize()
    assert_f16_equal(res, xs, "residual updated in place")
    assert np.array_equal(out.cpu().numpy(), q)
    assert_f16_equal(scale, s, "scale")
    assert_f16_equal(ssum, sm, "sum")


@pytest.mark.parametrize("tokens,d", [(16, 14336), (3, 28672), (5, 128), (1030, 14336), (128, 28672), (5, 32768), (2, 16416),
                                      (3, 32800)])
def test_fused_silu_mul_quant_matches_the_two_kernels(tokens, d):
    """Compared against the HIP silu_and_mul + quant pair (both go through the same device expf).  Rows of 16384 < d <= 32768
    (Llama-2-70B's 28672) take the <512, 8> geometry with 112 - 128 KiB of LDS, wider ones the one-kernel-per-row fallback."""
    import omniserve_backend.activation_ops as act
    import omniserve_backend.fused_kernels as fk
    from omniserve_amd.backend import fused_ext
    x = to_dev(_x(tokens, 2 * d, tokens + d, 2.0))
    tmp = torch.empty((tokens, d), dtype=torch.float16, device=dev())
    act.silu_and_mul(tmp, x)
    q1 = torch.empty((tokens, d), dtype=torch.int8, device=dev()); s1 = torch.empty((tokens,), dtype=torch.float16, device=dev()); m1 = s1.clone()
    fk.invoke_quant_fuse_sum(q1, tmp, m1, s1)
    q2 = torch.empty_like(q1); s2 = torch.empty_like(s1); m2 = torch.empty_like(s1)
    fused_ext.silu_mul_quant_fuse_sum(q2, x, m2, s2)
    torch.cuda.synchronize()
    assert torch.equal(q1, q2) and torch.equal(s1.view(torch.int16), s2.view(torch.int16)) and torch.equal(m1.view(torch.int16), m2.view(torch.int16))


@pytest.mark.parametrize("M,N,K", [(16, 4096, 4096), (16, 8192, 1024), (7, 5120, 2048)])
def test_deferred_splitk_epilogue_matches_gemm_then_add_norm(M, N, K):
    """o_proj / down_proj path of the fused runner: partial GEMM + slab-consuming add+norm must equal
    the reference sequence GEMM -> residual add -> rms_norm_general_fuse_sum bit for bit.  Rows of 4096 columns run one
    vector per thread, 4097 .. 8192 two with batched requests (Llama-2-70B's hidden size; round 4)."""
    import omniserve_backend.layernorm_ops as ln
    import omniserve_backend.qgemm_w4a8_per_chn as gemm
    from omniserve_amd.backend import fused_ext
    from oracle import w4a8
    u, z, s1 = w4a8.synth_per_channel(N, K, 3)
    qw, s1h, szh = w4a8.pack_per_channel(u, z, s1)
    a, sa, asum = oe.quant_per_token(_x(M, K, 5, 1.0), True)
    resid = _x(M, N, 6, 2.0)
    g = (1.0 + 0.1 * np.random.default_rng(1).standard_normal(N)).astype(np.float16)
    qw_d, s1_d, sz_d, a_d, sa_d, as_d, g_d = map(to_dev, (qw, s1h, szh, a, sa, asum, g))
    # reference sequence
    proj = torch.empty((M, N), dtype=torch.float16, device=dev())
    gemm.gemm_forward_cuda(a_d, qw_d, s1_d, sa_d, sz_d, as_d, proj)
    x1 = to_dev(resid); x1.add_(proj)
    q1 = torch.empty((M, N), dtype=torch.int8, device=dev()); sc1 = torch.empty((M,), dtype=torch.float16, device=dev()); sm1 = sc1.clone()
    ln.rms_norm_general_fuse_sum(q1, x1, g_d, sm1, sc1, 1e-5, True)
    # deferred epilogue
    slab = torch.empty((64 << 20,), dtype=torch.uint8, device=dev())
    sk = fused_ext.gemm_partial_per_chn(a_d, qw_d, slab)
    x2 = to_dev(resid)
    q2 = torch.empty_like(q1); sc2 = torch.empty_like(sc1); sm2 = torch.empty_like(sc1)
    fused_ext.splitk_add_rms_norm_general_fuse_sum(q2, x2, slab, sk, s1_d, sa_d, sz_d, as_d, g_d, sm2, sc2, 1e-5)
    torch.cuda.synchronize()
    assert sk >= 1
    assert torch.equal(x1.view(torch.int16), x2.view(torch.int16))
    assert torch.equal(q1, q2) and torch.equal(sc1.view(torch.int16), sc2.view(torch.int16))
    assert torch.equal(sm1.view(torch.int16), sm2.view(torch.int16))
    # and against the oracle
    want = w4a8.gemm_per_chn(a, qw, s1h, sa, szh, asum)
    xs = (resid.astype(np.float32) + want.astype(np.float32)).astype(np.float16)
    qo, so, smo = oe.rms_norm_general(xs, g, 1e-5, True)
    assert np.array_equal(q2.cpu().numpy(), qo)
    assert_f16_equal(sm2, smo, "sum")


@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (16, 4096, 14336), (3, 2048, 4096), (33, 4096, 4096)])
def test_deferred_splitk_w8_epilogue_matches_gemm_then_add_norm(M, N, K):
    """o_proj / down_proj of the fused LServe runner (W8A8): partial GEMM + slab-consuming add+norm must equal the
    reference sequence w8a8_gemm -> residual add -> rms_norm_general bit for bit (llama_w8a8_unpad.py:382-427)."""
    import omniserve_backend.layernorm_ops as ln
    import omniserve_backend.qgemm_w8a8 as gemm
    from omniserve_amd.backend import fused_ext
    from oracle import w4a8
    rng = np.random.default_rng(M + K)
    w = rng.integers(-128, 128, size=(N, K), dtype=np.int8)
    sw = rng.uniform(0.0002, 0.0014, size=(N,)).astype(np.float16)
    a, sa, _ = oe.quant_per_token(_x(M, K, 5, 1.0), False)
    resid = _x(M, N, 6, 2.0)
    g = (1.0 + 0.1 * np.random.default_rng(1).standard_normal(N)).astype(np.float16)
    w_d, sw_d, a_d, sa_d, g_d = map(to_dev, (w, sw, a, sa, g))
    # reference sequence
    proj = torch.empty((M, N), dtype=torch.float16, device=dev())
    gemm.w8a8_gemm_forward_cuda(a_d, w_d, sw_d, sa_d, proj)
    x1 = to_dev(resid); x1.add_(proj)
    q1 = torch.empty((M, N), dtype=torch.int8, device=dev()); sc1 = torch.empty((M,), dtype=torch.float16, device=dev())
    ln.rms_norm_general(q1, x1, g_d, sc1, 1e-5, True)
    # deferred epilogue
    slab = torch.empty((64 << 20,), dtype=torch.uint8, device=dev())
    sk = fused_ext.gemm_partial_w8a8(a_d, w_d, slab)
    x2 = to_dev(resid)
    q2 = torch.empty_like(q1); sc2 = torch.empty_like(sc1); sm2 = torch.empty_like(sc1)
    fused_ext.splitk_w8_add_rms_norm_general_fuse_sum(q2, x2, slab, sk, sw_d, sa_d, g_d, sm2, sc2, 1e-5)
    torch.cuda.synchronize()
    assert sk >= 1
    assert torch.equal(x1.view(torch.int16), x2.view(torch.int16))
    assert torch.equal(q1, q2) and torch.equal(sc1.view(torch.int16), sc2.view(torch.int16))
    # and against the oracle
    want = w4a8.gemm_w8a8(a, w, sw, sa)
    xs = (resid.astype(np.float32) + want.astype(np.float32)).astype(np.float16)
    qo, so, smo = oe.rms_norm_general(xs, g, 1e-5, True)
    assert np.array_equal(q2.cpu().numpy(), qo)
    assert_f16_equal(sc2, so, "scale")
    assert_f16_equal(sm2, smo, "sum")


@pytest.mark.parametrize("M,N,K", [(16, 4096, 4096), (64, 4096, 14336), (5, 1024, 2048)])
def test_deferred_splitk_per_group_epilogue_matches_gemm_then_add_norm(M, N, K):
    """g128 W4A8: partial GEMM + slab-consuming add+norm == per-group GEMM -> residual add -> rms_norm_general_fuse_sum, bit for bit
    (the per-group epilogue h(f32(acc) * (s1[n] * s_a[m])) is the one the W8A8 consumer applies)."""
    import omniserve_backend.layernorm_ops as ln
    import omniserve_backend.qgemm_w4a8_per_group as gemm
    from omniserve_amd.backend import fused_ext
    from oracle import w4a8
    u, z, s2, s1 = w4a8.synth_per_group(N, K, seed=M + 1)
    qw, s1h, s2s, s2z = w4a8.pack_per_group(u, z, s2, s1)
    a, sa, _ = oe.quant_per_token(_x(M, K, 5, 1.0), False)
    resid = _x(M, N, 6, 2.0)
    g = (1.0 + 0.1 * np.random.default_rng(1).standard_normal(N)).astype(np.float16)
    qw_d, s1_d, s2s_d, s2z_d, a_d, sa_d, g_d = map(to_dev, (qw, s1h, s2s, s2z, a, sa, g))
    proj = torch.empty((M, N), dtype=torch.float16, device=dev())
    gemm.gemm_forward_cuda(a_d, qw_d, s2z_d, s2s_d, s1_d, sa_d, proj)
    x1 = to_dev(resid); x1.add_(proj)
    q1 = torch.empty((M, N), dtype=torch.int8, device=dev()); sc1 = torch.empty((M,), dtype=torch.float16, device=dev()); sm1 = sc1.clone()
    ln.rms_norm_general_fuse_sum(q1, x1, g_d, sm1, sc1, 1e-5, True)
    slab = torch.empty((64 << 20,), dtype=torch.uint8, device=dev())
    sk = fused_ext.gemm_partial_per_group(a_d, qw_d, s2z_d, s2s_d, slab)
    x2 = to_dev(resid)
    q2 = torch.empty_like(q1); sc2 = torch.empty_like(sc1); sm2 = torch.empty_like(sc1)
    fused_ext.splitk_w8_add_rms_norm_general_fuse_sum(q2, x2, slab, sk, s1_d, sa_d, g_d, sm2, sc2, 1e-5)
    torch.cuda.synchronize()
    assert sk >= 1
    assert torch.equal(x1.view(torch.int16), x2.view(torch.int16))
    assert torch.equal(q1, q2) and torch.equal(sc1.view(torch.int16), sc2.view(torch.int16))
    assert torch.equal(sm1.view(torch.int16), sm2.view(torch.int16))
    want = w4a8.gemm_per_group(a, qw, s2z, s2s, s1h, sa)
    xs = (resid.astype(np.float32) + want.astype(np.float32)).astype(np.float16)
    qo, so, smo = oe.rms_norm_general(xs, g, 1e-5, True)
    assert np.array_equal(q2.cpu().numpy(), qo)
    assert_f16_equal(sm2, smo, "sum")


def test_argmax_matches_torch():
    """Greedy-sampling helper (fused_ext.argmax): first maximum, -0 == +0, NaN wins, ragged and unaligned shapes."""
    from omniserve_amd.backend import fused_ext
    g = torch.Generator(device="cpu").manual_seed(7)
    for rows, cols in ((16, 128256), (3, 1000), (1, 7), (5, 32000)):
        x = torch.randn((rows, cols), generator=g).half()
        x[0, cols // 2] = 9.0
        x[0, cols - 1] = 9.0                      # tie: the first index wins
        if rows > 1:
            x[1].zero_(); x[1, 3] = -0.0          # all equal (+0 / -0): index 0
        if rows > 2:
            x[2, min(500, cols - 1)] = float("nan")
        xd = x.to(dev())
        out = torch.empty((rows,), dtype=torch.int64, device=dev())
        fused_ext.argmax(out, xd)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), torch.argmax(x.float(), dim=-1)), (rows, cols)
    # strided rows (row stride not a multiple of 8 -> scalar path)
    big = torch.randn((4, 1003), generator=g).half().to(dev())
    view = big[:, :999]
    out = torch.empty((4,), dtype=torch.int64, device=dev())
    fused_ext.argmax(out, view)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), torch.argmax(view.float().cpu(), dim=-1))


def test_embed_rows_is_index_select():
    from omniserve_amd.backend import fused_ext
    g = torch.Generator(device="cpu").manual_seed(4)
    table = torch.randn((1000, 4096), generator=g).half().to(dev())
    idx = torch.randint(0, 1000, (16,), generator=g).to(dev())
    out = torch.zeros((16, 4096), dtype=torch.float16, device=dev())
    fused_ext.embed_rows(out, table, idx)
    torch.cuda.synchronize()
    assert torch.equal(out, torch.index_select(table, 0, idx))
    with pytest.raises(RuntimeError):
        fused_ext.embed_rows(out, table, idx.int())


def test_fused_entry_points_without_the_row_sum():
    """sum = None (W8A8 / per-group callers): same codes and scales as the summing form, nothing else written."""
    from omniserve_amd.backend import fused_ext
    rng = np.random.default_rng(12)
    T, H, I = 5, 4096, 14336
    g = (1.0 + 0.1 * rng.standard_normal(H)).astype(np.float16)
    res = (rng.standard_normal((T, H)) * 2).astype(np.float16)
    delta = rng.standard_normal((T, H)).astype(np.float16)
    outs = []
    for with_sum in (True, False):
        q = torch.empty((T, H), dtype=torch.int8, device=dev())
        sc = torch.empty((T,), dtype=torch.float16, device=dev())
        sm = torch.full((T,), 7.0, dtype=torch.float16, device=dev())
        r = to_dev(res.copy())
        fused_ext.add_rms_norm_general_fuse_sum(q, r, to_dev(delta), to_dev(g), sm if with_sum else None, sc, 1e-5)
        gu = to_dev((rng.standard_normal((T, 2 * I))).astype(np.float16)) if with_sum else gu
        q2 = torch.empty((T, I), dtype=torch.int8, device=dev())
        sc2 = torch.empty((T,), dtype=torch.float16, device=dev())
        fused_ext.silu_mul_quant_fuse_sum(q2, gu, sm if with_sum else None, sc2)
        torch.cuda.synchronize()
        outs.append((q.cpu().numpy(), sc.cpu().numpy(), r.cpu().numpy(), q2.cpu().numpy(), sc2.cpu().numpy()))
    for a, b in zip(*outs):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("flavour,M,N,K", [("chn", 16, 4096, 14336), ("chn", 3, 2048, 4096), ("w8", 1, 4096, 14336),
                                           ("w8", 16, 4096, 4096), ("grp", 16, 4096, 14336)])
def test_deferred_splitk_final_rms_norm_matches_gemm_then_add_norm(flavour, M, N, K):
    """The LAST layer's down projection consumed by the model's final norm: partial GEMM + splitk_add_rms_norm == GEMM ->
    residual add -> rms_norm (llama_w4a8_unpad.py:484 / llama_w8a8_unpad.py), bit for bit, and against the oracle."""
    import omniserve_backend.layernorm_ops as ln
    from omniserve_amd.backend import fused_ext
    from oracle import w4a8
    resid = _x(M, N, 6, 2.0)
    g = (1.0 + 0.1 * np.random.default_rng(1).standard_normal(N)).astype(np.float16)
    g_d = to_dev(g)
    proj = torch.empty((M, N), dtype=torch.float16, device=dev())
    slab = torch.empty((64 << 20,), dtype=torch.uint8, device=dev())
    if flavour == "chn":
        import omniserve_backend.qgemm_w4a8_per_chn as gemm
        u, z, s1 = w4a8.synth_per_channel(N, K, 3)
        qw, s1h, szh = w4a8.pack_per_channel(u, z, s1)
        a, sa, asum = oe.quant_per_token(_x(M, K, 5, 1.0), True)
        qw_d, s1_d, sz_d, a_d, sa_d, as_d = map(to_dev, (qw, s1h, szh, a, sa, asum))
        gemm.gemm_forward_cuda(a_d, qw_d, s1_d, sa_d, sz_d, as_d, proj)
        sk = fused_ext.gemm_partial_per_chn(a_d, qw_d, slab)
        args = (s1_d, sa_d, sz_d, as_d)
        want = w4a8.gemm_per_chn(a, qw, s1h, sa, szh, asum)
    elif flavour == "w8":
        import omniserve_backend.qgemm_w8a8 as gemm
        rng = np.random.default_rng(M + K)
        w = rng.integers(-128, 128, size=(N, K), dtype=np.int8)
        sw = rng.uniform(0.0002, 0.0014, size=(N,)).astype(np.float16)
        a, sa, _ = oe.quant_per_token(_x(M, K, 5, 1.0), False)
        w_d, sw_d, a_d, sa_d = map(to_dev, (w, sw, a, sa))
        gemm.w8a8_gemm_forward_cuda(a_d, w_d, sw_d, sa_d, proj)
        sk = fused_ext.gemm_partial_w8a8(a_d, w_d, slab)
        args = (sw_d, sa_d, None, None)
        want = w4a8.gemm_w8a8(a, w, sw, sa)
    else:
        import omniserve_backend.qgemm_w4a8_per_group as gemm
        u, z, s2, s1 = w4a8.synth_per_group(N, K, seed=M + 1)
        qw, s1h, s2s, s2z = w4a8.pack_per_group(u, z, s2, s1)
        a, sa, _ = oe.quant_per_token(_x(M, K, 5, 1.0), False)
        qw_d, s1_d, s2s_d, s2z_d, a_d, sa_d = map(to_dev, (qw, s1h, s2s, s2z, a, sa))
        gemm.gemm_forward_cuda(a_d, qw_d, s2z_d, s2s_d, s1_d, sa_d, proj)
        sk = fused_ext.gemm_partial_per_group(a_d, qw_d, s2z_d, s2s_d, slab)
        args = (s1_d, sa_d, None, None)
        want = w4a8.gemm_per_group(a, qw, s2z, s2s, s1h, sa)
    x1 = to_dev(resid); x1.add_(proj)
    o1 = torch.empty((M, N), dtype=torch.float16, device=dev())
    ln.rms_norm(o1, x1, g_d, 1e-5, False)
    x2 = to_dev(resid)
    o2 = torch.full((M, N), 7.0, dtype=torch.float16, device=dev())
    fused_ext.splitk_add_rms_norm(o2, x2, slab, sk, *args, g_d, 1e-5)
    torch.cuda.synchronize()
    assert sk >= 1
    assert torch.equal(x1.view(torch.int16), x2.view(torch.int16)), "residual"
    assert torch.equal(o1.view(torch.int16), o2.view(torch.int16)), "final norm output"
    xs = (resid.astype(np.float32) + want.astype(np.float32)).astype(np.float16)
    assert_f16_equal(x2, xs, "residual vs oracle")
    assert_f16_equal(o2, oe.rms_norm(xs, g, 1e-5), "rms_norm vs oracle")


def test_decode_step_begin_is_three_torch_ops():
    from omniserve_amd.backend import fused_ext
    g = torch.Generator(device="cpu").manual_seed(5)
    table = torch.randn((1000, 4096), generator=g).half().to(dev())
    idx = torch.randint(0, 1000, (16,), generator=g).to(dev())
    lengths = torch.arange(100, 116, dtype=torch.int32, device=dev())
    amax = torch.full((32, 2, 1024), 0x3F800000, dtype=torch.int32, device=dev())
    out = torch.zeros((16, 4096), dtype=torch.float16, device=dev())
    fused_ext.decode_step_begin(out, table, idx, lengths, amax)
    torch.cuda.synchronize()
    assert torch.equal(out, torch.index_select(table, 0, idx))
    assert torch.equal(lengths.cpu(), torch.arange(101, 117, dtype=torch.int32))
    assert int(amax.abs().sum()) == 0
    # optional parts; a bad id fills its row with NaN as embed_rows does
    idx[3] = 5000
    fused_ext.decode_step_begin(out, table, idx)
    torch.cuda.synchronize()
    assert torch.isnan(out[3].float()).all() and torch.equal(out[4], table[idx[4]])
    assert torch.equal(lengths.cpu(), torch.arange(101, 117, dtype=torch.int32))
