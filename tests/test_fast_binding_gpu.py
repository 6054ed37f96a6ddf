"""The pybind11 fast path of the mirror (omniserve_amd/csrc_ext/omni_ext.cpp) against the ctypes path: the same C-ABI calls, so
the same bits, the same in-place / view semantics and the same error type -- for every mirror function that routes to it."""
import numpy as np
import pytest
import torch

from omniserve_amd import _lib
from tests.util import dev

pytestmark = pytest.mark.gpu


class _Path:
    def __init__(self, use_ext):
        self.use_ext = use_ext

    def __enter__(self):
        self.keep = _lib.USE_EXT
        _lib.USE_EXT = self.use_ext
        assert (_lib.fast() is not None) == self.use_ext, "the fast path must be built (python -m omniserve_amd.build)"

    def __exit__(self, *exc):
        _lib.USE_EXT = self.keep


def _both(fn):
    out = []
    for use_ext in (False, True):
        with _Path(use_ext):
            out.append(fn())
    torch.cuda.synchronize()
    return out


def _eq(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x.dtype == y.dtype and torch.equal(x.view(torch.uint8) if x.dtype != torch.float16 else x.view(torch.int16),
                                                  y.view(torch.uint8) if y.dtype != torch.float16 else y.view(torch.int16))


@pytest.mark.parametrize("M,N,K", [(16, 4096, 4096), (16, 4096, 14336), (64, 512, 2048), (128, 28672, 4096), (300, 512, 1024)])
def test_gemms_agree(M, N, K):
    import omniserve_backend.qgemm_w4a8_per_chn as chn
    import omniserve_backend.qgemm_w4a8_per_group as grp
    import omniserve_backend.qgemm_w8a8 as w8
    g = torch.Generator(device=dev()).manual_seed(M + N)
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev(), generator=g)
    w = torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev(), generator=g).view(torch.int8)
    w8w = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev(), generator=g)
    sw = (torch.rand((N,), device=dev(), generator=g) * 0.01 + 0.002).half()
    sz = (sw.float() * 7).half()
    sa = (torch.rand((M,), device=dev(), generator=g) * 0.01 + 0.002).half()
    asum = (torch.randn((M,), device=dev(), generator=g) * 3).half()
    s2s = torch.randint(1, 9, (K // 128, N), dtype=torch.int8, device=dev(), generator=g)
    s2z = torch.randint(-100, 1, (K // 128, N), dtype=torch.int8, device=dev(), generator=g)

    def run():
        buf = torch.full((M + 2, N), 7.0, dtype=torch.float16, device=dev())
        o1, o2, o3 = buf[1:M + 1], torch.empty((M, N), dtype=torch.float16, device=dev()), torch.empty((M, N), dtype=torch.float16, device=dev())
        assert chn.gemm_forward_cuda(a, w, sw, sa, sz, asum, o1) is None       # in place into a row-slice view
        grp.gemm_forward_cuda(a, w, s2z, s2s, sw, sa, o2)
        w8.w8a8_gemm_forward_cuda(a, w8w, sw, sa, o3)
        return buf, o2, o3
    r0, r1 = _both(run)
    _eq(r0, r1)
    assert (r1[0][0] == 7).all() and (r1[0][M + 1] == 7).all()


@pytest.mark.parametrize("tokens,hidden", [(16, 4096), (1, 4096), (300, 5120), (16, 14336)])
def test_row_kernels_agree(tokens, hidden):
    import omniserve_backend.activation_ops as act
    import omniserve_backend.fused_kernels as fk
    import omniserve_backend.layernorm_ops as ln
    g = torch.Generator(device=dev()).manual_seed(tokens + hidden)
    x = torch.randn((tokens, hidden), dtype=torch.float32, device=dev(), generator=g).half()
    gamma = (torch.rand((hidden,), device=dev(), generator=g) + 0.5).half()
    gu = torch.randn((tokens, 2 * hidden), dtype=torch.float32, device=dev(), generator=g).half()

    def run():
        q1, q2, q3, q4 = (torch.empty((tokens, hidden), dtype=torch.int8, device=dev()) for _ in range(4))
        s = [torch.empty((tokens,), dtype=torch.float16, device=dev()) for _ in range(6)]
        y = torch.empty((tokens, hidden), dtype=torch.float16, device=dev())
        o = torch.empty((tokens, hidden), dtype=torch.float16, device=dev())
        ln.rms_norm_general_fuse_sum(q1, x, gamma, s[0], s[1], 1e-5, True)
        ln.rms_norm_general(q2, x, gamma, s[2], 1e-5, True)
        ln.rms_norm(y, x, gamma, 1e-5)
        fk.invoke_quant_fuse_sum(q3, x, s[3], s[4])
        fk.invoke_quant(q4, x, s[5])
        act.silu_and_mul(o, gu)
        return [q1, q2, q3, q4, y, o] + s
    r0, r1 = _both(run)
    _eq(r0, r1)


def test_errors_are_runtime_errors_on_both_paths():
    import omniserve_backend.fused_kernels as fk
    import omniserve_backend.qgemm_w4a8_per_chn as chn
    import omniserve_backend.qgemm_w4a8_per_group as grp
    M, N, K = 16, 256, 256
    a = torch.zeros((M, K), dtype=torch.int8, device=dev())
    w = torch.zeros((N, K // 2), dtype=torch.int8, device=dev())
    h = torch.zeros((max(M, N),), dtype=torch.float16, device=dev())
    out = torch.empty((M, N), dtype=torch.float16, device=dev())
    for use_ext in (False, True):
        with _Path(use_ext):
            with pytest.raises(RuntimeError):       # host tensor: no CPU path
                chn.gemm_forward_cuda(a.cpu(), w, h[:N], h[:M], h[:N], h[:M], out)
            with pytest.raises(RuntimeError):       # fp16 activations
                chn.gemm_forward_cuda(a.half(), w, h[:N], h[:M], h[:N], h[:M], out)
            with pytest.raises(RuntimeError):       # weight shape
                chn.gemm_forward_cuda(a, w[:, :-1].contiguous(), h[:N], h[:M], h[:N], h[:M], out)
            with pytest.raises(RuntimeError):       # second-level parameters of the wrong shape
                grp.gemm_forward_cuda(a, w, torch.zeros((1, N), dtype=torch.int8, device=dev()),
                                      torch.zeros((K // 128, N), dtype=torch.int8, device=dev()), h[:N], h[:M], out)
            with pytest.raises(RuntimeError):       # non-contiguous quantiser input
                x = torch.zeros((M, 2 * K), dtype=torch.float16, device=dev())[:, ::2]
                fk.invoke_quant(torch.empty((M, K), dtype=torch.int8, device=dev()), x, h[:M])
            fk.invoke_quant(torch.empty((0, K), dtype=torch.int8, device=dev()), torch.empty((0, K), dtype=torch.float16, device=dev()),
                            torch.empty((0,), dtype=torch.float16, device=dev()))     # zero rows: a no-op
    torch.cuda.synchronize()


def test_other_element_types_keep_the_ctypes_path():
    """bf16 / fp32 rows and the static-scale overloads are not in the fast module: they still work with it enabled."""
    import omniserve_backend.fused_kernels as fk
    x = torch.randn((4, 256), dtype=torch.float32, device=dev()).to(torch.bfloat16)
    q = torch.empty((4, 256), dtype=torch.int8, device=dev())
    s = torch.empty((4,), dtype=torch.float16, device=dev())
    with _Path(True):
        fk.invoke_quant(q, x, s)
        fk.invoke_quant(q, x.half(), 0.05)
    torch.cuda.synchronize()
    assert np.isfinite(s.float().cpu().numpy()).all()


def test_decode_attention_agrees_incl_appended_rows():
    """fused_attention_pure_dense.single_query_attention: callee-allocated result and the appended KV rows, both bindings."""
    import omniserve_backend.fused_attention_pure_dense as fa
    from oracle import kv4
    from tests.util import GpuPagedKV, to_dev
    D, BASE, Hq, Hk = 128, 500000.0, 32, 8
    hist = [200, 17, 130, 1000]
    B = len(hist)
    rng = np.random.default_rng(7)
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
    pools = [GpuPagedKV(kc, vc, kidx, vidx), GpuPagedKV(kc, vc, kidx, vidx)]
    lens = to_dev(np.asarray(hist, np.int32) + 1)
    qkv = to_dev(rng.standard_normal((B, (Hq + 2 * Hk) * D)).astype(np.float16))
    q = qkv[:, : Hq * D].view(B, Hq, D)
    k = qkv[:, Hq * D:(Hq + Hk) * D].view(B, Hk, D)
    v = qkv[:, (Hq + Hk) * D:].view(B, Hk, D)
    outs = []
    for use_ext, g in zip((False, True), pools):
        with _Path(use_ext):
            outs.append(fa.single_query_attention(q, k, v, g.table, lens, None, 65536, 64, Hk * D // 2, max(hist) + 1, D, BASE, True, True, True))
    torch.cuda.synchronize()
    assert outs[1].shape == (B, Hq, D) and outs[1].is_contiguous()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    assert torch.equal(pools[0].kpool, pools[1].kpool) and torch.equal(pools[0].vpool, pools[1].vpool)
    with _Path(True):
        with pytest.raises(RuntimeError):
            fa.single_query_attention(q, k, v, pools[1].table, lens.long(), None, 65536, 64, Hk * D // 2, max(hist) + 1, D, BASE, True, True, True)
        with pytest.raises(NotImplementedError):      # non-neox RoPE: refused before either binding is reached
            fa.single_query_attention(q, k, v, pools[1].table, lens, None, 65536, 64, Hk * D // 2, max(hist) + 1, D, BASE, False, True, True)
