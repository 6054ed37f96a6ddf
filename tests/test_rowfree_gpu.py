"""Round-3 fused extension: the decode layer without its quantiser row kernels.  Every new entry point against the
reference call sequence it replaces (HIP) AND against the oracle, bit for bit:

  gemm_silu_*                == gemm_forward_cuda -> silu_and_mul   (fp16 activation) + row maxima
  gemm_partial_f16_*         == invoke_quant_fuse_sum -> gemm partial (slabs, sums, scales)
  decode_attention_f16_amax  == single_query_attention              (fp16 output) + row maxima
"""
import numpy as np
import pytest
import torch

from oracle import elementwise as oe
from oracle import w4a8
from tests.util import assert_f16_equal, dev, to_dev

pytestmark = pytest.mark.gpu


def _x(tokens, hidden, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((tokens, hidden)) * scale).astype(np.float16)


def _row_amax(slots, rows):
    """max over the candidates of each row, as float32 (f32 bit patterns of non-negative values, [XCD][row][sub])."""
    return slots.cpu().numpy().view(np.float32).reshape(8, 16, 8).max(axis=(0, 2))[:rows]


@pytest.mark.parametrize("M,N,K", [(16, 28672, 4096), (1, 28672, 4096), (7, 2048, 512), (5, 256, 128),
                                   (16, 1024, 960), (8, 512, 1152)])   # (the last two: one / two K parts per workgroup + leftover steps)
def test_gemm_silu_per_chn(M, N, K):
    import omniserve_backend.activation_ops as act_ops
    import omniserve_backend.qgemm_w4a8_per_chn as gemm
    from omniserve_amd.backend import fused_ext
    u, z, s1 = w4a8.synth_per_channel(N, K, M + N)
    qw, s1h, szh = w4a8.pack_per_channel(u, z, s1)
    a, sa, asum = oe.quant_per_token(_x(M, K, 5 + M, 1.0), True)
    qw_d, s1_d, sz_d, a_d, sa_d, as_d = map(to_dev, (qw, s1h, szh, a, sa, asum))
    # reference sequence on the device
    gu = torch.empty((M, N), dtype=torch.float16, device=dev())
    gemm.gemm_forward_cuda(a_d, qw_d, s1_d, sa_d, sz_d, as_d, gu)
    want_act = torch.empty((M, N // 2), dtype=torch.float16, device=dev())
    act_ops.silu_and_mul(want_act, gu)
    # fused
    act = torch.full((M, N // 2), 7.0, dtype=torch.float16, device=dev())
    amax = fused_ext.new_amax_slots(M, dev())
    fused_ext.gemm_silu_per_chn(a_d, qw_d, s1_d, sa_d, sz_d, as_d, act, amax)
    torch.cuda.synchronize()
    assert torch.equal(act.view(torch.int16), want_act.view(torch.int16))
    assert np.array_equal(_row_amax(amax, M), want_act.float().abs().max(dim=1).values.cpu().numpy())
    # the GEMM half against the oracle (silu goes through the device's exp: compared HIP to HIP above)
    assert_f16_equal(gu, w4a8.gemm_per_chn(a, qw, s1h, sa, szh, asum), "gate_up GEMM vs oracle")
    # raising is monotone: a second call on the same slots cannot lower them, a call with larger inputs raises them
    before = _row_amax(amax, M).copy()
    fused_ext.gemm_silu_per_chn(a_d, qw_d, s1_d, sa_d, sz_d, as_d, act, amax)
    torch.cuda.synchronize()
    assert np.array_equal(_row_amax(amax, M), before)


def test_gemm_silu_rejects_shapes_it_does_not_cover():
    """M > 16 and K that needs a grid-level split are refused (the caller keeps the two-kernel sequence), never
    computed wrongly."""
    from omniserve_amd.backend import fused_ext
    for M, N, K in ((17, 1024, 512), (4, 1024, 8192)):
        a = torch.zeros((M, K), dtype=torch.int8, device=dev())
        w = torch.zeros((N, K // 2), dtype=torch.int8, device=dev())
        h = lambda n: torch.zeros((n,), dtype=torch.float16, device=dev())  # noqa: E731
        with pytest.raises(RuntimeError):
            fused_ext.gemm_silu_per_chn(a, w, h(N), h(M), h(N), h(M), torch.empty((M, N // 2), dtype=torch.float16, device=dev()),
                                        fused_ext.new_amax_slots(M, dev()))


@pytest.mark.parametrize("M,N,K", [(16, 28672, 4096), (3, 2048, 512), (16, 1024, 1024)])
def test_gemm_silu_per_group(M, N, K):
    import omniserve_backend.activation_ops as act_ops
    import omniserve_backend.qgemm_w4a8_per_group as gemm
    from omniserve_amd.backend import fused_ext
    u, z, s2, s1 = w4a8.synth_per_group(N, K, seed=M + 2)
    qw, s1h, s2s, s2z = w4a8.pack_per_group(u, z, s2, s1)
    a, sa, _ = oe.quant_per_token(_x(M, K, 9, 1.0), False)
    qw_d, s1_d, s2s_d, s2z_d, a_d, sa_d = map(to_dev, (qw, s1h, s2s, s2z, a, sa))
    gu = torch.empty((M, N), dtype=torch.float16, device=dev())
    gemm.gemm_forward_cuda(a_d, qw_d, s2z_d, s2s_d, s1_d, sa_d, gu)
    want_act = torch.empty((M, N // 2), dtype=torch.float16, device=dev())
    act_ops.silu_and_mul(want_act, gu)
    act = torch.empty((M, N // 2), dtype=torch.float16, device=dev())
    amax = fused_ext.new_amax_slots(M, dev())
    fused_ext.gemm_silu_per_group(a_d, qw_d, s2z_d, s2s_d, s1_d, sa_d, act, amax)
    torch.cuda.synchronize()
    assert torch.equal(act.view(torch.int16), want_act.view(torch.int16))
    assert np.array_equal(_row_amax(amax, M), want_act.float().abs().max(dim=1).values.cpu().numpy())


def _amax_from(act_np, M, spread_seed):
    """Slots as a producer could have left them: the row maximum in one slot, smaller candidates / zeros in the others."""
    rng = np.random.default_rng(spread_seed)
    mx = np.abs(act_np.astype(np.float32)).max(axis=1)
    slots = np.zeros((8, 16, 8), np.float32)
    for m in range(M):
        slots[:, m, :] = mx[m] * rng.random((8, 8)) * (rng.random((8, 8)) > 0.3)
        slots[rng.integers(0, 8), m, rng.integers(0, 8)] = mx[m]
    slots[:, M:, :] = 1e30          # rows the activation does not have: must never be read into a result
    return to_dev(slots.reshape(-1).view(np.int32))


@pytest.mark.parametrize("M,N,K", [(16, 4096, 14336), (16, 4096, 4096), (1, 4096, 14336), (5, 512, 1024), (4, 512, 512),
                                   (16, 4096, 16384), (9, 1024, 4096 + 512)])
@pytest.mark.parametrize("edge", [False, True])
def test_gemm_partial_f16_per_chn(M, N, K, edge):
    """Slabs == those of invoke_quant_fuse_sum -> gemm_partial_per_chn; sums / scales == invoke_quant_fuse_sum's (also
    vs the oracle).  edge: an all-zero row (amax 0: 127/0 = inf, codes 0, scale 0), a +-65504 row, a row whose sum
    overflows fp16 (Appendix A.2 and friends, through the on-the-fly quantiser)."""
    import omniserve_backend.fused_kernels as fk
    from omniserve_amd.backend import fused_ext
    u, z, s1 = w4a8.synth_per_channel(N, K, 3 + M)
    qw, _, _ = w4a8.pack_per_channel(u, z, s1)
    x = _x(M, K, 11 + K, 2.0)
    if edge:
        x[0] = 0
        if M > 2:
            x[1, 0::2] = np.float16(65504.0); x[1, 1::2] = np.float16(-65504.0)
            x[2] = np.float16(900.0)
    x_d, qw_d = to_dev(x), to_dev(qw)
    slab1 = torch.empty((64 << 20,), dtype=torch.uint8, device=dev())
    slab2 = torch.zeros((64 << 20,), dtype=torch.uint8, device=dev())
    q = torch.empty((M, K), dtype=torch.int8, device=dev())
    sc1 = torch.empty((M,), dtype=torch.float16, device=dev()); sm1 = sc1.clone()
    fk.invoke_quant_fuse_sum(q, x_d, sm1, sc1)
    sk1 = fused_ext.gemm_partial_per_chn(q, qw_d, slab1)
    sc2 = torch.full((M,), -1.0, dtype=torch.float16, device=dev()); sm2 = sc2.clone()
    sk2 = fused_ext.gemm_partial_f16_per_chn(x_d, _amax_from(x, M, K), qw_d, slab2, sm2, sc2)
    torch.cuda.synchronize()
    assert sk1 == sk2 >= 1
    n = sk1 * M * N * 4
    assert torch.equal(slab1[:n], slab2[:n]), "int32 slabs differ"
    assert torch.equal(sc1.view(torch.int16), sc2.view(torch.int16)), "scales"
    a = sm1.view(torch.int16).cpu().numpy(); b = sm2.view(torch.int16).cpu().numpy()
    nan = (a & 0x7FFF) > 0x7C00
    assert np.array_equal(nan, (b & 0x7FFF) > 0x7C00) and np.array_equal(a[~nan], b[~nan]), "sums"
    with np.errstate(all="ignore"):
        qo, so, smo = oe.quant_per_token(x, True)
    assert np.array_equal(q.cpu().numpy(), qo)
    ok = ~np.isnan(smo.astype(np.float32))
    assert np.array_equal(sm2.cpu().numpy().view(np.uint16)[ok], smo.view(np.uint16)[ok])
    assert np.array_equal(sc2.cpu().numpy().view(np.uint16), so.view(np.uint16))
    # the sum of the slabs is the integer GEMM of the oracle's codes
    acc = slab2[:n].view(torch.int32).view(sk2, M, N).sum(dim=0).cpu().numpy()
    want = qo.astype(np.int32) @ u.astype(np.int32).T
    assert np.array_equal(acc, want)


@pytest.mark.parametrize("M,N,K", [(16, 4096, 14336), (3, 512, 1024)])
def test_gemm_partial_f16_per_group(M, N, K):
    import omniserve_backend.fused_kernels as fk
    from omniserve_amd.backend import fused_ext
    u, z, s2, s1 = w4a8.synth_per_group(N, K, seed=M + 5)
    qw, s1h, s2s, s2z = w4a8.pack_per_group(u, z, s2, s1)
    x = _x(M, K, 13, 2.0)
    x_d, qw_d, s2s_d, s2z_d = map(to_dev, (x, qw, s2s, s2z))
    slab1 = torch.empty((64 << 20,), dtype=torch.uint8, device=dev())
    slab2 = torch.zeros((64 << 20,), dtype=torch.uint8, device=dev())
    q = torch.empty((M, K), dtype=torch.int8, device=dev())
    sc1 = torch.empty((M,), dtype=torch.float16, device=dev())
    fk.invoke_quant(q, x_d, sc1)
    sk1 = fused_ext.gemm_partial_per_group(q, qw_d, s2z_d, s2s_d, slab1)
    sc2 = torch.full((M,), -1.0, dtype=torch.float16, device=dev())
    sk2 = fused_ext.gemm_partial_f16_per_group(x_d, _amax_from(x, M, 3), qw_d, s2z_d, s2s_d, slab2, None, sc2)
    torch.cuda.synchronize()
    assert sk1 == sk2 >= 1
    n = sk1 * M * N * 4
    assert torch.equal(slab1[:n], slab2[:n])
    assert torch.equal(sc1.view(torch.int16), sc2.view(torch.int16))


@pytest.mark.parametrize("hist,Hq,Hk", [([200, 17, 130, 1], 32, 8), ([1500, 1030], 32, 8), ([63, 64, 65], 8, 2),
                                         ([90, 33], 4, 1)])
@pytest.mark.parametrize("single_launch", [True, False])
def test_decode_attention_f16_amax(hist, Hq, Hk, single_launch):
    """fp16 output == single_query_attention's, KV pages byte-identical, row maxima == max |out| per sequence.
    single_launch: the splits are merged by the last-arriving workgroup inside the attention launch (round 4) -- run three
    times in a row on the same ticket words (they must come back to zero), with an armed L2 prefetch riding along."""
    import omniserve_backend.fused_attention_pure_dense as fa
    from omniserve_amd.backend import fused_ext
    from oracle import kv4
    from tests.util import GpuPagedKV
    D, BASE = 128, 500000.0
    rng = np.random.default_rng(sum(hist) + Hq)
    B = len(hist)
    pages = (max(hist) + 64) // 64 + 1
    n_pages = B * pages
    kc, vc = kv4.PagedKV4(n_pages, Hk, D), kv4.PagedKV4(n_pages, Hk, D)
    for c in (kc, vc):
        c.pool[:] = rng.integers(0, 256, c.pool.shape, dtype=np.uint8)
        for p in range(n_pages):
            c.scales(p)[:] = (0.05 + 0.15 * rng.random((Hk, 64))).astype(np.float16)
            c.zeros(p)[:] = (6.0 + 3.0 * rng.random((Hk, 64))).astype(np.float16)
    kidx = rng.permutation(n_pages).reshape(B, pages)
    vidx = rng.permutation(n_pages).reshape(B, pages)
    g1, g2 = GpuPagedKV(kc, vc, kidx, vidx), GpuPagedKV(kc, vc, kidx, vidx)
    lens = to_dev(np.asarray(hist, np.int32) + 1)
    qkv = to_dev(rng.standard_normal((B, (Hq + 2 * Hk) * D)).astype(np.float16))
    q = qkv[:, : Hq * D].view(B, Hq, D)
    k = qkv[:, Hq * D:(Hq + Hk) * D].view(B, Hk, D)
    v = qkv[:, (Hq + Hk) * D:].view(B, Hk, D)
    T = max(hist) + 1
    want = fa.single_query_attention(q, k, v, g1.table, lens, None, 65536, 64, Hk * D // 2, T, D, BASE, True, True, True)
    out = torch.empty((B, Hq * D), dtype=torch.float16, device=dev())
    amax = fused_ext.new_amax_slots(B, dev())
    fused_ext.decode_attention_f16_amax(out, amax, q, k, v, g2.table, lens, 64, T, BASE, single_launch=single_launch)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), want.reshape(B, Hq * D).view(torch.int16))
    assert np.array_equal(_row_amax(amax, B), want.reshape(B, -1).float().abs().max(dim=1).values.cpu().numpy())
    for a, b in zip(g1.pools(), g2.pools()):
        assert np.array_equal(a, b)
    if single_launch:      # again (the appended row is rewritten with the same bytes), tickets re-used, riders on the grid
        wq = torch.zeros((4096, 2048), dtype=torch.int8, device=dev())
        for rep in range(3):
            out.fill_(7.0)
            amax.zero_()
            fused_ext.prefetch_arm_gemm(wq, B, 4096, 4096, 0, True, 8 << 20, 160)
            fused_ext.decode_attention_f16_amax(out, amax, q, k, v, g2.table, lens, 64, T, BASE)
            torch.cuda.synchronize()
            assert torch.equal(out.view(torch.int16), want.reshape(B, Hq * D).view(torch.int16)), rep
            assert np.array_equal(_row_amax(amax, B), want.reshape(B, -1).float().abs().max(dim=1).values.cpu().numpy())
        assert not fused_ext._tickets(dev()).any()


@pytest.mark.parametrize("nsplit", [3, 5, 6, 10])
def test_decode_attention_single_launch_fresh_inputs_stale_lines(nsplit):
    """The in-launch merge reads other workgroups' partials of the SAME launch (agent-scope stores, ticket, agent-scope loads:
    csrc/row_kernels.h SrcAttnMergeT<true>).  Every repeat has NEW q / k / v and NEW page contents, is compared with a fresh
    two-launch result, and runs right after (a) the two-launch path has left the PREVIOUS repeat's partials in the same
    workspace and (b) a reduction over the whole workspace has pulled those lines into the L1s / L2s of the chip: a merger
    that hit a stale line would reproduce the previous repeat's values.  nsplit % 4 != 0 makes neighbouring ticket groups
    share 128-B lines of the (m, l) array (8 B per (head, split))."""
    import omniserve_backend.fused_attention_pure_dense as fa  # noqa: F401  (page layout helpers live with the mirrors)
    from omniserve_amd import _lib
    from omniserve_amd.backend import fused_ext
    from oracle import kv4
    from tests.util import GpuPagedKV
    D, BASE, Hq, Hk = 128, 500000.0, 32, 8
    hist = [1000, 900, 1023, 700, 650, 1010]
    B = len(hist)
    rng = np.random.default_rng(nsplit)
    pages = (max(hist) + 64) // 64 + 1
    n_pages = B * pages
    kc, vc = kv4.PagedKV4(n_pages, Hk, D), kv4.PagedKV4(n_pages, Hk, D)
    for c in (kc, vc):
        c.pool[:] = rng.integers(0, 256, c.pool.shape, dtype=np.uint8)
        for p in range(n_pages):
            c.scales(p)[:] = (0.05 + 0.15 * rng.random((Hk, 64))).astype(np.float16)
            c.zeros(p)[:] = (6.0 + 3.0 * rng.random((Hk, 64))).astype(np.float16)
    kidx = rng.permutation(n_pages).reshape(B, pages)
    vidx = rng.permutation(n_pages).reshape(B, pages)
    g1, g2 = GpuPagedKV(kc, vc, kidx, vidx), GpuPagedKV(kc, vc, kidx, vidx)
    data_bytes = Hk * 64 * D // 2
    lens = to_dev(np.asarray(hist, np.int32) + 1)
    T = max(hist) + 1
    gen = torch.Generator(device=dev()).manual_seed(100 + nsplit)
    lib = _lib.lib()
    ws = _lib.workspace(lib.omni_kv4_decode_workspace_bytes(B, Hq, D, T), dev(), "attn")
    lib.omni_kv4_decode_set_split_override(nsplit)
    try:
        for rep in range(24):
            qkv = torch.randn((B, (Hq + 2 * Hk) * D), generator=gen, device=dev(), dtype=torch.float32).half() * (1.0 + rep % 3)
            q = qkv[:, : Hq * D].view(B, Hq, D)
            k = qkv[:, Hq * D:(Hq + Hk) * D].view(B, Hk, D)
            v = qkv[:, (Hq + Hk) * D:].view(B, Hk, D)
            for a, b in ((g1.kpool, g2.kpool), (g1.vpool, g2.vpool)):
                a[:, :data_bytes] = torch.randint(0, 256, (a.shape[0], data_bytes), generator=gen, device=dev(), dtype=torch.uint8)
                b.copy_(a)
            want = torch.empty((B, Hq * D), dtype=torch.float16, device=dev())
            amax1, amax2 = fused_ext.new_amax_slots(B, dev()), fused_ext.new_amax_slots(B, dev())
            fused_ext.decode_attention_f16_amax(want, amax1, q, k, v, g1.table, lens, 64, T, BASE, single_launch=False)
            _ = float(ws.view(torch.int32).sum())          # every CU reads the workspace: the two-launch partials are cached
            out = torch.full((B, Hq * D), 7.0, dtype=torch.float16, device=dev())
            fused_ext.decode_attention_f16_amax(out, amax2, q, k, v, g2.table, lens, 64, T, BASE, single_launch=True)
            torch.cuda.synchronize()
            assert torch.equal(out.view(torch.int16), want.view(torch.int16)), (nsplit, rep)
            assert np.array_equal(_row_amax(amax1, B), _row_amax(amax2, B)), (nsplit, rep)
            assert torch.equal(g1.kpool, g2.kpool) and torch.equal(g1.vpool, g2.vpool)
        assert not fused_ext._tickets(dev()).any()
    finally:
        lib.omni_kv4_decode_set_split_override(0)


# ---- W8A8 forms (LServe models): same hand-off, no zero-point term, no row sums ---------------------------------------
def _w8(N, K, seed):
    rng = np.random.default_rng(seed)
    w = rng.integers(-128, 128, size=(N, K), dtype=np.int8)
    sw = rng.uniform(0.001, 0.01, size=(N,)).astype(np.float16)
    return w, sw


@pytest.mark.parametrize("M,N,K", [(1, 28672, 4096), (16, 28672, 4096), (7, 2048, 512), (5, 256, 128), (16, 16384, 960),
                                   (3, 16384, 2048)])   # (256 x 128: leftover steps only; 960: ring rounds + leftovers)
def test_gemm_silu_w8a8(M, N, K):
    """act == w8a8_gemm_forward_cuda -> silu_and_mul bit for bit; row maxima == max |act|; GEMM half vs the oracle."""
    import omniserve_backend.activation_ops as act_ops
    import omniserve_backend.qgemm_w8a8 as gemm
    from omniserve_amd.backend import fused_ext
    w, sw = _w8(N, K, M + N)
    a, sa, _ = oe.quant_per_token(_x(M, K, 5 + M, 1.0), False)
    w_d, sw_d, a_d, sa_d = map(to_dev, (w, sw, a, sa))
    gu = torch.empty((M, N), dtype=torch.float16, device=dev())
    gemm.w8a8_gemm_forward_cuda(a_d, w_d, sw_d, sa_d, gu)
    want_act = torch.empty((M, N // 2), dtype=torch.float16, device=dev())
    act_ops.silu_and_mul(want_act, gu)
    act = torch.full((M, N // 2), 7.0, dtype=torch.float16, device=dev())
    amax = fused_ext.new_amax_slots(M, dev())
    fused_ext.gemm_silu_w8a8(a_d, w_d, sw_d, sa_d, act, amax)
    torch.cuda.synchronize()
    assert torch.equal(act.view(torch.int16), want_act.view(torch.int16))
    assert np.array_equal(_row_amax(amax, M), want_act.float().abs().max(dim=1).values.cpu().numpy())
    assert_f16_equal(gu, w4a8.gemm_w8a8(a, w, sw, sa), "gate_up W8A8 GEMM vs oracle")


def test_gemm_silu_w8a8_rejects_a_plan_with_a_grid_level_split():
    from omniserve_amd.backend import fused_ext
    M, N, K = 4, 1024, 8192
    a = torch.zeros((M, K), dtype=torch.int8, device=dev())
    w = torch.zeros((N, K), dtype=torch.int8, device=dev())
    h = lambda n: torch.ones((n,), dtype=torch.float16, device=dev())  # noqa: E731
    with pytest.raises(RuntimeError):
        fused_ext.gemm_silu_w8a8(a, w, h(N), h(M), torch.empty((M, N // 2), dtype=torch.float16, device=dev()),
                                 fused_ext.new_amax_slots(M, dev()))


@pytest.mark.parametrize("M,N,K", [(1, 4096, 14336), (16, 4096, 14336), (1, 4096, 4096), (16, 4096, 4096), (5, 512, 1024),
                                   (4, 512, 512), (9, 1024, 4096 + 512), (2, 1024, 28672)])
@pytest.mark.parametrize("edge", [False, True])
def test_gemm_partial_f16_w8a8(M, N, K, edge):
    """Slabs == invoke_quant -> gemm_partial_w8a8; scales == invoke_quant's (also vs the oracle); K beyond the int4 forms'
    16384 limit (no ordered row sum to stage).  edge: all-zero row, +-65504 row, constant row."""
    import omniserve_backend.fused_kernels as fk
    from omniserve_amd.backend import fused_ext
    w, _ = _w8(N, K, 3 + M)
    x = _x(M, K, 11 + K, 2.0)
    if edge:
        x[0] = 0
        if M > 2:
            x[1, 0::2] = np.float16(65504.0); x[1, 1::2] = np.float16(-65504.0)
            x[2] = np.float16(900.0)
    x_d, w_d = to_dev(x), to_dev(w)
    slab1 = torch.empty((64 << 20,), dtype=torch.uint8, device=dev())
    slab2 = torch.zeros((64 << 20,), dtype=torch.uint8, device=dev())
    q = torch.empty((M, K), dtype=torch.int8, device=dev())
    sc1 = torch.empty((M,), dtype=torch.float16, device=dev())
    fk.invoke_quant(q, x_d, sc1)
    sk1 = fused_ext.gemm_partial_w8a8(q, w_d, slab1)
    sc2 = torch.full((M,), -1.0, dtype=torch.float16, device=dev())
    sk2 = fused_ext.gemm_partial_f16_w8a8(x_d, _amax_from(x, M, K), w_d, slab2, sc2)
    torch.cuda.synchronize()
    assert sk1 == sk2 >= 1
    n = sk1 * M * N * 4
    assert torch.equal(slab1[:n], slab2[:n]), "int32 slabs differ"
    assert torch.equal(sc1.view(torch.int16), sc2.view(torch.int16)), "scales"
    with np.errstate(all="ignore"):
        qo, so, _ = oe.quant_per_token(x, False)
    assert np.array_equal(q.cpu().numpy(), qo)
    assert np.array_equal(sc2.cpu().numpy().view(np.uint16), so.view(np.uint16))
    acc = slab2[:n].view(torch.int32).view(sk2, M, N).sum(dim=0).cpu().numpy()
    assert np.array_equal(acc, qo.astype(np.int32) @ w.astype(np.int32).T)


# ---- q / k / v of the current token straight from the qkv projection's split-K slabs ---------------------------------------
def _kv_pools(B, hist, Hk, rng):
    from oracle import kv4
    from tests.util import GpuPagedKV
    D = 128
    pages = (max(hist) + 64) // 64 + 1
    n_pages = B * pages
    kc, vc = kv4.PagedKV4(n_pages, Hk, D), kv4.PagedKV4(n_pages, Hk, D)
    for c in (kc, vc):
        c.pool[:] = rng.integers(0, 256, c.pool.shape, dtype=np.uint8)
        for p in range(n_pages):
            c.scales(p)[:] = (0.05 + 0.15 * rng.random((Hk, 64))).astype(np.float16)
            c.zeros(p)[:] = (6.0 + 3.0 * rng.random((Hk, 64))).astype(np.float16)
    kidx = rng.permutation(n_pages).reshape(B, pages)
    vidx = rng.permutation(n_pages).reshape(B, pages)
    return GpuPagedKV(kc, vc, kidx, vidx), GpuPagedKV(kc, vc, kidx, vidx)


@pytest.mark.parametrize("flavour,hist,Hq,Hk,K", [("chn", [200, 17, 130, 1, 700], 32, 8, 4096), ("chn", [63, 64, 65], 8, 2, 512),
                                                  ("grp", [1500, 1030], 32, 8, 4096), ("w8", [90, 33, 300], 4, 1, 2048)])
def test_decode_attention_reads_qkv_from_the_projections_slabs(flavour, hist, Hq, Hk, K):
    """qkv GEMM -> single_query_attention   ==   qkv partial GEMM (int32 slabs) -> decode_arm_qkv_slabs -> attention:
    output and KV pages byte-identical; the fp16 q / k / v buffer is not read (filled with NaN)."""
    import omniserve_backend.fused_attention_pure_dense as fa
    from omniserve_amd.backend import fused_ext
    D, BASE = 128, 500000.0
    rng = np.random.default_rng(sum(hist) + Hq + K)
    B = len(hist)
    N = (Hq + 2 * Hk) * D
    g1, g2 = _kv_pools(B, hist, Hk, rng)
    lens = to_dev(np.asarray(hist, np.int32) + 1)
    slab = torch.empty((64 << 20,), dtype=torch.uint8, device=dev())
    qkv = torch.empty((B, N), dtype=torch.float16, device=dev())
    if flavour == "chn":
        import omniserve_backend.qgemm_w4a8_per_chn as gemm
        u, z, s1 = w4a8.synth_per_channel(N, K, 3)
        qw, s1h, szh = w4a8.pack_per_channel(u, z, s1)
        a, sa, asum = oe.quant_per_token(_x(B, K, 5, 1.0), True)
        qw_d, s1_d, sz_d, a_d, sa_d, as_d = map(to_dev, (qw, s1h, szh, a, sa, asum))
        gemm.gemm_forward_cuda(a_d, qw_d, s1_d, sa_d, sz_d, as_d, qkv)
        sk = fused_ext.gemm_partial_per_chn(a_d, qw_d, slab)
        epi = (s1_d, sa_d, sz_d, as_d)
    elif flavour == "grp":
        import omniserve_backend.qgemm_w4a8_per_group as gemm
        u, z, s2, s1 = w4a8.synth_per_group(N, K, seed=4)
        qw, s1h, s2s, s2z = w4a8.pack_per_group(u, z, s2, s1)
        a, sa, _ = oe.quant_per_token(_x(B, K, 5, 1.0), False)
        qw_d, s1_d, s2s_d, s2z_d, a_d, sa_d = map(to_dev, (qw, s1h, s2s, s2z, a, sa))
        gemm.gemm_forward_cuda(a_d, qw_d, s2z_d, s2s_d, s1_d, sa_d, qkv)
        sk = fused_ext.gemm_partial_per_group(a_d, qw_d, s2z_d, s2s_d, slab)
        epi = (s1_d, sa_d, None, None)
    else:
        import omniserve_backend.qgemm_w8a8 as gemm
        w, sw = _w8(N, K, 6)
        a, sa, _ = oe.quant_per_token(_x(B, K, 5, 1.0), False)
        w_d, sw_d, a_d, sa_d = map(to_dev, (w, sw, a, sa))
        gemm.w8a8_gemm_forward_cuda(a_d, w_d, sw_d, sa_d, qkv)
        sk = fused_ext.gemm_partial_w8a8(a_d, w_d, slab)
        epi = (sw_d, sa_d, None, None)
    views = lambda t: (t[:, : Hq * D].view(B, Hq, D), t[:, Hq * D:(Hq + Hk) * D].view(B, Hk, D),  # noqa: E731
                       t[:, (Hq + Hk) * D:].view(B, Hk, D))
    T = max(hist) + 1
    q, k, v = views(qkv)
    want = fa.single_query_attention(q, k, v, g1.table, lens, None, 65536, 64, Hk * D // 2, T, D, BASE, True, True, True)
    junk = torch.full((B, N), float("nan"), dtype=torch.float16, device=dev())
    q2, k2, v2 = views(junk)
    fused_ext.decode_arm_qkv_slabs(slab, sk, B, N, 0, Hq * D, (Hq + Hk) * D, *epi)
    got = fa.single_query_attention(q2, k2, v2, g2.table, lens, None, 65536, 64, Hk * D // 2, T, D, BASE, True, True, True)
    torch.cuda.synchronize()
    assert sk >= 1
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    for x, y in zip(g1.pools(), g2.pools()):
        assert np.array_equal(x, y)
    # one shot: the next call reads its fp16 arguments again
    g3, _ = _kv_pools(B, hist, Hk, np.random.default_rng(sum(hist) + Hq + K))
    again = fa.single_query_attention(q, k, v, g3.table, lens, None, 65536, 64, Hk * D // 2, T, D, BASE, True, True, True)
    torch.cuda.synchronize()
    assert torch.equal(again.view(torch.int16), want.view(torch.int16))
