"""Round-4 fused extension: the MLP half of the decode layer as ONE persistent launch (csrc/mlp_fused.hip:
omni_w4a8_per_chn_mlp_fused) against the three-launch sequence it replaces, bit for bit --

    splitk_add_rms_norm_general_fuse_sum  ->  gemm_silu_per_chn  ->  gemm_partial_f16_per_chn

(each of which tests/test_elementwise_gpu.py and tests/test_rowfree_gpu.py pin to the oracle) -- and the quantities the
oracle can state directly: the residual stream, the int8 codes' GEMM (sum of the down_proj slabs = A_q . U^T), the row sums.
The in-kernel hand-offs are exercised the way they run in a decode step: several launches ("layers") in a row on counters
zeroed once, different data every launch (a stale line of the previous launch's scratch would change bits), with a
concurrent bandwidth hog on a second stream in one case (uneven arrival)."""
import numpy as np
import pytest
import torch

from oracle import elementwise as oe
from oracle import w4a8
from tests.util import dev, to_dev

pytestmark = pytest.mark.gpu

H = 4096


def _layer_inputs(M, inter, seed, sk_o=3):
    rng = np.random.default_rng(seed)
    d = {}
    d["x"] = (rng.standard_normal((M, H)) * 0.7).astype(np.float16)
    d["o_slab"] = rng.integers(-40000, 40000, size=(sk_o, M, H), dtype=np.int32)
    d["o_ws"] = (0.002 + 0.004 * rng.random(H)).astype(np.float16)
    d["o_wsz"] = (d["o_ws"].astype(np.float32) * rng.integers(0, 16, H)).astype(np.float16)
    d["o_as"] = (0.005 + 0.01 * rng.random(M)).astype(np.float16)
    d["o_asum"] = (rng.standard_normal(M) * 3.0).astype(np.float16)
    d["gamma"] = (1.0 + 0.05 * rng.standard_normal(H)).astype(np.float16)
    d["Wgu"] = rng.integers(0, 256, size=(2 * inter, H // 2), dtype=np.uint8).view(np.int8)
    d["gu_ws"] = (0.002 + 0.018 * rng.random(2 * inter)).astype(np.float16)
    d["gu_wsz"] = (d["gu_ws"].astype(np.float32) * rng.integers(0, 16, 2 * inter)).astype(np.float16)
    d["Wdn"] = rng.integers(0, 256, size=(H, inter // 2), dtype=np.uint8).view(np.int8)
    return d


def _reference(dd, M, inter, eps):
    """The level-3 launch sequence on the device -> (residual, down slabs [sk, M, H], act sums, act scales, fp16 activation)."""
    from omniserve_amd.backend import fused_ext
    x = dd["x"].clone()
    q = torch.empty((M, H), dtype=torch.int8, device=dev())
    sB = torch.empty((M,), dtype=torch.float16, device=dev()); mB = torch.empty_like(sB)
    fused_ext.splitk_add_rms_norm_general_fuse_sum(q, x, dd["o_slab"], dd["o_slab"].shape[0], dd["o_ws"], dd["o_as"], dd["o_wsz"],
                                                   dd["o_asum"], dd["gamma"], mB, sB, eps)
    act = torch.empty((M, inter), dtype=torch.float16, device=dev())
    amax = fused_ext.new_amax_slots(M, dev())
    fused_ext.gemm_silu_per_chn(q, dd["Wgu"], dd["gu_ws"], sB, dd["gu_wsz"], mB, act, amax)
    slab = torch.zeros((16, M, H), dtype=torch.int32, device=dev())
    s2 = torch.empty((M,), dtype=torch.float16, device=dev()); m2 = torch.empty_like(s2)
    sk = fused_ext.gemm_partial_f16_per_chn(act, amax, dd["Wdn"], slab, m2, s2)
    torch.cuda.synchronize()
    return x, slab[:sk].clone(), m2, s2, act


def _fused(dd, M, inter, eps, counters, scratch, layers, phase, clocks=False):
    from omniserve_amd.backend import fused_ext
    x = dd["x"].clone()
    slab = torch.zeros((16, M, H), dtype=torch.int32, device=dev())
    s2 = torch.full((M,), 9.0, dtype=torch.float16, device=dev()); m2 = torch.full_like(s2, 9.0)
    sk = fused_ext.mlp_fused_per_chn(x, dd["o_slab"], dd["o_slab"].shape[0], dd["o_ws"], dd["o_wsz"], dd["o_as"], dd["o_asum"],
                                     dd["gamma"], eps, dd["Wgu"], dd["gu_ws"], dd["gu_wsz"], dd["Wdn"], slab, m2, s2, counters,
                                     layers, phase, scratch, clocks=clocks)
    return x, slab, sk, m2, s2


def _same_bits(a, b):
    return torch.equal(a.contiguous().view(torch.int16), b.contiguous().view(torch.int16))


@pytest.mark.parametrize("M,inter", [(16, 14336), (16, 8192), (5, 14336), (1, 10240)])
def test_mlp_fused_matches_the_three_launch_sequence(M, inter):
    from omniserve_amd.backend import fused_ext
    if not fused_ext.mlp_fused_ok(M, H, inter):
        pytest.skip("the persistent MLP launch needs 256 CUs")
    eps, layers = 1e-5, 4
    counters, scratch = fused_ext.mlp_fused_buffers(layers, H, inter, dev())
    for phase in range(layers):                      # one zeroing of the counters, four launches with different data
        host = _layer_inputs(M, inter, 100 * M + inter + phase)
        dd = {k: to_dev(v) for k, v in host.items()}
        want_x, want_slab, want_m2, want_s2, want_act = _reference(dd, M, inter, eps)
        x, slab, sk, m2, s2 = _fused(dd, M, inter, eps, counters, scratch, layers, phase)
        torch.cuda.synchronize()
        fused_ext.mlp_fused_check(counters)
        assert sk == want_slab.shape[0] == inter // 2048
        assert _same_bits(x, want_x), "residual stream, launch %d" % phase
        assert torch.equal(slab[:sk], want_slab), "down_proj slabs, launch %d" % phase
        assert not slab[sk:].any()
        assert _same_bits(m2, want_m2) and _same_bits(s2, want_s2), "activation row sums / scales, launch %d" % phase
        if phase == 0:
            # against the oracle directly: the residual stream (o_proj's epilogue on the slabs' sum, fp16 add), the quantiser
            # of the activation, and the codes' GEMM through down_proj (sum of the slabs = A_q . U^T)
            acc = host["o_slab"].sum(axis=0).astype(np.float32)
            t = ((acc * host["o_ws"].astype(np.float32)[None, :]).astype(np.float32) * host["o_as"].astype(np.float32)[:, None]).astype(np.float32)
            c = (host["o_wsz"].astype(np.float32)[None, :] * host["o_asum"].astype(np.float32)[:, None]).astype(np.float32)
            ep = (t - c).astype(np.float32).astype(np.float16)           # = oracle.w4a8.gemm_per_chn's epilogue
            xo = (host["x"].astype(np.float32) + ep.astype(np.float32)).astype(np.float16)
            assert np.array_equal(x.cpu().numpy().view(np.uint16), xo.view(np.uint16)), "residual vs oracle"
            a_q, a_s, a_sum = oe.quant_per_token(want_act.cpu().numpy(), True)
            assert np.array_equal(slab[:sk].sum(dim=0).cpu().numpy(), w4a8.gemm_per_chn_acc(a_q, host["Wdn"])), "sum of slabs vs oracle GEMM"
            assert np.array_equal(s2.cpu().numpy().view(np.uint16), a_s.view(np.uint16)), "scales vs oracle quantiser"
            assert np.array_equal(m2.cpu().numpy().view(np.uint16), a_sum.view(np.uint16)), "row sums vs oracle quantiser"


def test_mlp_fused_under_uneven_load_and_replay():
    """The hand-offs with a bandwidth hog on a second stream (workgroups arrive unevenly), captured in a HIP graph and
    replayed (counters re-zeroed by a memset node at the head of the graph, as the decode step does once per step)."""
    from omniserve_amd.backend import fused_ext
    M, inter, eps, layers = 16, 14336, 1e-5, 3
    if not fused_ext.mlp_fused_ok(M, H, inter):
        pytest.skip("the persistent MLP launch needs 256 CUs")
    counters, scratch = fused_ext.mlp_fused_buffers(layers, H, inter, dev())
    dds = [{k: to_dev(v) for k, v in _layer_inputs(M, inter, 7 + p).items()} for p in range(layers)]
    wants = [_reference(dd, M, inter, eps) for dd in dds]
    xs = [dd["x"].clone() for dd in dds]
    slabs = [torch.zeros((16, M, H), dtype=torch.int32, device=dev()) for _ in range(layers)]
    m2 = [torch.empty((M,), dtype=torch.float16, device=dev()) for _ in range(layers)]
    s2 = [torch.empty((M,), dtype=torch.float16, device=dev()) for _ in range(layers)]

    def step():
        counters.zero_()
        for p, dd in enumerate(dds):
            xs[p].copy_(dd["x"])
            fused_ext.mlp_fused_per_chn(xs[p], dd["o_slab"], 3, dd["o_ws"], dd["o_wsz"], dd["o_as"], dd["o_asum"], dd["gamma"], eps,
                                        dd["Wgu"], dd["gu_ws"], dd["gu_wsz"], dd["Wdn"], slabs[p], m2[p], s2[p], counters, layers,
                                        p, scratch)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    hog_src = torch.empty((256 << 20,), dtype=torch.uint8, device=dev())
    hog_dst = torch.empty_like(hog_src)
    hog = torch.cuda.Stream()
    for rep in range(6):
        for s in slabs:
            s.zero_()
        torch.cuda.synchronize()
        if rep % 2:
            with torch.cuda.stream(hog):
                for _ in range(4):
                    hog_dst.copy_(hog_src)
        g.replay()
        torch.cuda.synchronize()
        fused_ext.mlp_fused_check(counters)
        for p in range(layers):
            want_x, want_slab, want_m2, want_s2, _ = wants[p]
            assert _same_bits(xs[p], want_x), (rep, p)
            assert torch.equal(slabs[p][:7], want_slab), (rep, p)
            assert _same_bits(m2[p], want_m2) and _same_bits(s2[p], want_s2), (rep, p)


def test_mlp_fused_rejects_layers_it_does_not_cover():
    from omniserve_amd.backend import fused_ext
    assert not fused_ext.mlp_fused_ok(17, 4096, 14336)
    assert not fused_ext.mlp_fused_ok(16, 5120, 13824)
    assert not fused_ext.mlp_fused_ok(16, 4096, 28672)      # more than two units per workgroup
    assert not fused_ext.mlp_fused_ok(16, 4096, 4096)
