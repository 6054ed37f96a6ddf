"""Round-6 fused extension: a (residual add + norm + quant) -> (decode GEMV) pair of the decoder layer as ONE launch
(csrc/norm_gemv_fused.h; llama_w4a8_unpad.py:410-432).  Against the call sequence it replaces on the device AND the oracle,
bit for bit: residual, int8 codes, scales, sums, projection output (or SiLU activation + row maxima).

Hand-off stress: fresh inputs every repeat with the previous repeat's codes / pairs planted in the caches chip-wide, and the
rows delayed behind a bandwidth hog on another stream (late producer) -- a consumer that read a stale line or passed the gate
early would differ from the two-launch reference.
"""
import numpy as np
import pytest
import torch

from oracle import elementwise as oe
from oracle import w4a8
from tests.util import assert_f16_equal, dev, to_dev

pytestmark = pytest.mark.gpu


class _Lin:
    def __init__(self, qw, s1, sz=None, s2z=None, s2s=None):
        self.qweight, self.s1_scales = qw, s1
        self.group = -1 if sz is not None else 128
        if sz is not None:
            self.s1_szeros = sz
        else:
            self.s2_zeros, self.s2_scales = s2z, s2s


def _x(tokens, hidden, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((tokens, hidden)) * scale).astype(np.float16)


def _row_amax(slots, rows):
    return slots.cpu().numpy().view(np.float32).reshape(8, 16, 8).max(axis=(0, 2))[:rows]


def _sync(n=1):
    from omniserve_amd.backend import fused_ext
    return (torch.zeros((n, fused_ext.NGF_SYNC_WORDS), dtype=torch.int32, device=dev()),
            torch.zeros((4,), dtype=torch.int32, device=dev()))


def _producer_slabs(M, H, Kp, seed):
    """Split-K slabs of a per-channel producer GEMM (o_proj / down_proj) with its scales, as the decode step leaves them."""
    from omniserve_amd.backend import fused_ext
    from omniserve_amd import _lib
    u, z, s1 = w4a8.synth_per_channel(H, Kp, seed)
    qw, s1h, szh = w4a8.pack_per_channel(u, z, s1)
    a, sa, asum = oe.quant_per_token(_x(M, Kp, seed + 1, 1.0), True)
    qw_d, s1_d, sz_d, a_d, sa_d, as_d = map(to_dev, (qw, s1h, szh, a, sa, asum))
    need = int(_lib.lib().omni_gemm_partial_workspace_bytes(M, H, Kp))
    slab = torch.empty((max(need, 1 << 20),), dtype=torch.uint8, device=dev())
    sk = fused_ext.gemm_partial_per_chn(a_d, qw_d, slab)
    return _Lin(qw_d, s1_d, sz_d), slab, sk, sa_d, as_d


@pytest.mark.parametrize("M,N,H,Kp,silu", [(16, 6144, 4096, 14336, False), (16, 28672, 4096, 4096, True), (1, 6144, 4096, 4096, False),
                                            (7, 2048, 512, 1024, True), (5, 256, 256, 512, False), (16, 1024, 2048, 2048, True),
                                            (3, 512, 1024, 1024, False)])
def test_norm_gemm_fused_slab_source_per_chn(M, N, H, Kp, silu):
    """rows from split-K slabs (the decode layer's form): == splitk_add_rms_norm_general_fuse_sum -> gemm[_silu]."""
    import omniserve_backend.qgemm_w4a8_per_chn as gemm
    from omniserve_amd.backend import fused_ext
    assert fused_ext.norm_gemm_fused_ok(M, N, H, -1, silu)
    prod, slab, sk, p_sa, p_as = _producer_slabs(M, H, Kp, 100 + M)
    u, z, s1 = w4a8.synth_per_channel(N, H, 7 + M)
    qw, s1h, szh = w4a8.pack_per_channel(u, z, s1)
    lin = _Lin(*map(to_dev, (qw, s1h, szh)))
    gamma = to_dev((1.0 + 0.05 * np.random.default_rng(3).standard_normal(H)).astype(np.float16))
    res0 = to_dev(_x(M, H, 21, 1.0))
    h = lambda *s: torch.empty(s, dtype=torch.float16, device=dev())  # noqa: E731
    # ---- the call sequence it replaces
    res_w = res0.clone()
    codes_w = torch.empty((M, H), dtype=torch.int8, device=dev())
    sum_w, scale_w = h(M), h(M)
    fused_ext.splitk_add_rms_norm_general_fuse_sum(codes_w, res_w, slab, sk, prod.s1_scales, p_sa, prod.s1_szeros, p_as, gamma,
                                                   sum_w, scale_w, 1e-5)
    if silu:
        out_w = h(M, N // 2)
        amax_w = fused_ext.new_amax_slots(M, dev())
        fused_ext.gemm_silu_per_chn(codes_w, lin.qweight, lin.s1_scales, scale_w, lin.s1_szeros, sum_w, out_w, amax_w)
    else:
        out_w = h(M, N)
        gemm.gemm_forward_cuda(codes_w, lin.qweight, lin.s1_scales, scale_w, lin.s1_szeros, sum_w, out_w)
    # ---- one launch
    res = res0.clone()
    codes = torch.full((M, H), 77, dtype=torch.int8, device=dev())
    sum_f, scale_f = h(M), h(M)
    out = torch.full_like(out_w, 3.0)
    amax = fused_ext.new_amax_slots(M, dev()) if silu else None
    sync, err = _sync()
    fused_ext.norm_gemm_fused(codes, res, gamma, sum_f, scale_f, 1e-5, lin, out, sync[0], err, slab=slab, sk=sk, producer=prod,
                              p_ascales=p_sa, p_asums=p_as, amax=amax)
    torch.cuda.synchronize()
    assert int(err[0].item()) == 0
    assert torch.equal(res.view(torch.int16), res_w.view(torch.int16)), "residual"
    assert torch.equal(codes, codes_w), "codes"
    assert torch.equal(scale_f.view(torch.int16), scale_w.view(torch.int16)), "scales"
    assert torch.equal(sum_f.view(torch.int16), sum_w.view(torch.int16)), "sums"
    assert torch.equal(out.view(torch.int16), out_w.view(torch.int16)), "projection"
    if silu:
        assert np.array_equal(_row_amax(amax, M), _row_amax(amax_w, M))
    else:   # the GEMM half against the oracle
        assert_f16_equal(out, w4a8.gemm_per_chn(codes_w.cpu().numpy(), qw, s1h, scale_w.cpu().numpy(), szh, sum_w.cpu().numpy()),
                         "fused projection vs oracle")
    # the rows against the oracle
    want_q, want_s, want_sum = oe.rms_norm_general(res_w.cpu().numpy(), gamma.cpu().numpy(), 1e-5, True)
    assert np.array_equal(codes.cpu().numpy(), want_q)
    assert np.array_equal(scale_f.cpu().numpy().view(np.int16), want_s.view(np.int16))
    assert np.array_equal(sum_f.cpu().numpy().view(np.int16), want_sum.view(np.int16))


@pytest.mark.parametrize("src", ["plain", "delta"])
@pytest.mark.parametrize("M,N,H", [(16, 6144, 4096), (4, 512, 512)])
def test_norm_gemm_fused_other_sources_per_chn(src, M, N, H):
    import omniserve_backend.layernorm_ops as ln
    import omniserve_backend.qgemm_w4a8_per_chn as gemm
    from omniserve_amd.backend import fused_ext
    u, z, s1 = w4a8.synth_per_channel(N, H, 5)
    lin = _Lin(*map(to_dev, w4a8.pack_per_channel(u, z, s1)))
    gamma = to_dev((1.0 + 0.05 * np.random.default_rng(4).standard_normal(H)).astype(np.float16))
    res0, delta = to_dev(_x(M, H, 31, 1.0)), to_dev(_x(M, H, 32, 0.5))
    h = lambda *s: torch.empty(s, dtype=torch.float16, device=dev())  # noqa: E731
    res_w, codes_w, sum_w, scale_w, out_w = res0.clone(), torch.empty((M, H), dtype=torch.int8, device=dev()), h(M), h(M), h(M, N)
    if src == "delta":
        fused_ext.add_rms_norm_general_fuse_sum(codes_w, res_w, delta, gamma, sum_w, scale_w, 1e-5)
    else:
        ln.rms_norm_general_fuse_sum(codes_w, res_w, gamma, sum_w, scale_w, 1e-5, True)
    gemm.gemm_forward_cuda(codes_w, lin.qweight, lin.s1_scales, scale_w, lin.s1_szeros, sum_w, out_w)
    res, codes, sum_f, scale_f, out = res0.clone(), torch.empty((M, H), dtype=torch.int8, device=dev()), h(M), h(M), h(M, N)
    sync, err = _sync()
    fused_ext.norm_gemm_fused(codes, res, gamma, sum_f, scale_f, 1e-5, lin, out, sync[0], err, delta=delta if src == "delta" else None)
    torch.cuda.synchronize()
    assert int(err[0].item()) == 0
    for got, want, what in ((res, res_w, "residual"), (scale_f, scale_w, "scales"), (sum_f, sum_w, "sums"), (out, out_w, "projection")):
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), what
    assert torch.equal(codes, codes_w)


@pytest.mark.parametrize("M,N,H,silu", [(16, 28672, 4096, True), (16, 6144, 4096, False), (3, 1024, 512, True)])
def test_norm_gemm_fused_per_group(M, N, H, silu):
    """g128 layers: rms_norm_general (no row sum) -> per-group GEMM; rows from a per-group producer's slabs."""
    import omniserve_backend.qgemm_w4a8_per_group as gemm
    from omniserve_amd import _lib
    from omniserve_amd.backend import fused_ext
    assert fused_ext.norm_gemm_fused_ok(M, N, H, 128, silu)
    Kp = 1024
    pu, pz, ps2, ps1 = w4a8.synth_per_group(H, Kp, seed=11)
    pqw, ps1h, ps2s, ps2z = map(to_dev, w4a8.pack_per_group(pu, pz, ps2, ps1))
    pa, psa, _ = oe.quant_per_token(_x(M, Kp, 12, 1.0), False)
    pa_d, psa_d = to_dev(pa), to_dev(psa)
    slab = torch.empty((max(int(_lib.lib().omni_gemm_partial_workspace_bytes(M, H, Kp)), 1 << 20),), dtype=torch.uint8, device=dev())
    sk = fused_ext.gemm_partial_per_group(pa_d, pqw, ps2z, ps2s, slab)
    prod = _Lin(pqw, ps1h, None, ps2z, ps2s)
    u, z, s2, s1 = w4a8.synth_per_group(N, H, seed=M + 2)
    qw, s1h, s2s, s2z = map(to_dev, w4a8.pack_per_group(u, z, s2, s1))
    lin = _Lin(qw, s1h, None, s2z, s2s)
    gamma = to_dev((1.0 + 0.05 * np.random.default_rng(6).standard_normal(H)).astype(np.float16))
    res0 = to_dev(_x(M, H, 41, 1.0))
    h = lambda *s: torch.empty(s, dtype=torch.float16, device=dev())  # noqa: E731
    res_w, codes_w, scale_w = res0.clone(), torch.empty((M, H), dtype=torch.int8, device=dev()), h(M)
    fused_ext.splitk_w8_add_rms_norm_general_fuse_sum(codes_w, res_w, slab, sk, prod.s1_scales, psa_d, gamma, None, scale_w, 1e-5)
    if silu:
        out_w, amax_w = h(M, N // 2), fused_ext.new_amax_slots(M, dev())
        fused_ext.gemm_silu_per_group(codes_w, qw, s2z, s2s, s1h, scale_w, out_w, amax_w)
    else:
        out_w = h(M, N)
        gemm.gemm_forward_cuda(codes_w, qw, s2z, s2s, s1h, scale_w, out_w)
    res, codes, scale_f, out = res0.clone(), torch.empty((M, H), dtype=torch.int8, device=dev()), h(M), torch.empty_like(out_w)
    amax = fused_ext.new_amax_slots(M, dev()) if silu else None
    sync, err = _sync()
    fused_ext.norm_gemm_fused(codes, res, gamma, None, scale_f, 1e-5, lin, out, sync[0], err, slab=slab, sk=sk, producer=prod,
                              p_ascales=psa_d, amax=amax)
    torch.cuda.synchronize()
    assert int(err[0].item()) == 0
    assert torch.equal(res.view(torch.int16), res_w.view(torch.int16))
    assert torch.equal(codes, codes_w)
    assert torch.equal(scale_f.view(torch.int16), scale_w.view(torch.int16))
    assert torch.equal(out.view(torch.int16), out_w.view(torch.int16))
    if silu:
        assert np.array_equal(_row_amax(amax, M), _row_amax(amax_w, M))


def test_norm_gemm_fused_refuses_what_it_does_not_cover():
    from omniserve_amd.backend import fused_ext
    assert not fused_ext.norm_gemm_fused_ok(17, 6144, 4096)          # more than 16 rows
    assert not fused_ext.norm_gemm_fused_ok(16, 6144, 8192)          # a K part beyond the register ring
    assert not fused_ext.norm_gemm_fused_ok(16, 6144, 4096 + 64)     # K parts of whole k-steps
    assert not fused_ext.norm_gemm_fused_ok(16, 64 * 1024, 4096)     # the grid (1040 workgroups) would not be resident at once
    assert not fused_ext.norm_gemm_fused_ok(16, 6144 + 64, 4096, -1, True)


def test_norm_gemm_fused_hand_off_under_stale_lines_and_late_rows():
    """24 repeats on the decode shapes with FRESH residuals / slabs each time, all into the SAME code / pair / sync buffers: before
    every fused launch the previous repeat's codes and pairs are read by a chip-wide kernel (planting those lines in every
    L2 and many L1s) and a bandwidth hog runs on a second stream, so rows publish late and unevenly.  Every word of the outputs
    must equal the two-launch sequence's."""
    import omniserve_backend.qgemm_w4a8_per_chn as gemm
    from omniserve_amd.backend import fused_ext
    M, N, H, Kp = 16, 6144, 4096, 4096
    u, z, s1 = w4a8.synth_per_channel(N, H, 9)
    lin = _Lin(*map(to_dev, w4a8.pack_per_channel(u, z, s1)))
    gamma = to_dev((1.0 + 0.05 * np.random.default_rng(8).standard_normal(H)).astype(np.float16))
    h = lambda *s: torch.empty(s, dtype=torch.float16, device=dev())  # noqa: E731
    codes, sum_f, scale_f, out = torch.zeros((M, H), dtype=torch.int8, device=dev()), h(M), h(M), h(M, N)
    sync, err = _sync()
    hog_src = torch.empty((256 << 20,), dtype=torch.uint8, device=dev())
    hog_dst = torch.empty_like(hog_src)
    side = torch.cuda.Stream()
    for rep in range(24):
        prod, slab, sk, p_sa, p_as = _producer_slabs(M, H, Kp, 1000 + rep)
        res0 = to_dev(_x(M, H, 2000 + rep, 1.0 + 0.1 * rep))
        res_w, codes_w, sum_w, scale_w, out_w = res0.clone(), torch.empty_like(codes), h(M), h(M), h(M, N)
        fused_ext.splitk_add_rms_norm_general_fuse_sum(codes_w, res_w, slab, sk, prod.s1_scales, p_sa, prod.s1_szeros, p_as, gamma,
                                                       sum_w, scale_w, 1e-5)
        gemm.gemm_forward_cuda(codes_w, lin.qweight, lin.s1_scales, scale_w, lin.s1_szeros, sum_w, out_w)
        # plant the previous repeat's lines: every CU reads the old codes / pairs (a big elementwise kernel over them, repeated)
        junk = codes.view(torch.int32).repeat(64, 1).sum() + sync.sum()
        sync.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(3):
                hog_dst.copy_(hog_src)
        res = res0.clone()
        fused_ext.norm_gemm_fused(codes, res, gamma, sum_f, scale_f, 1e-5, lin, out, sync[0], err, slab=slab, sk=sk, producer=prod,
                                  p_ascales=p_sa, p_asums=p_as)
        torch.cuda.synchronize()
        assert int(err[0].item()) == 0 and junk is not None
        assert torch.equal(codes, codes_w), "codes, repeat %d" % rep
        assert torch.equal(res.view(torch.int16), res_w.view(torch.int16)), "residual, repeat %d" % rep
        assert torch.equal(sum_f.view(torch.int16), sum_w.view(torch.int16)) and torch.equal(scale_f.view(torch.int16), scale_w.view(torch.int16))
        assert torch.equal(out.view(torch.int16), out_w.view(torch.int16)), "projection, repeat %d" % rep
