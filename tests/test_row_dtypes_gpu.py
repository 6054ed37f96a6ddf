"""bf16 / fp32 element types of the row kernels (the reference dispatches invoke_quant[_fuse_sum], rms_norm,
rms_norm_general[_fuse_sum] and silu_and_mul over float, half and bfloat16: kernels/csrc/dispatch_utils.h:7-14) against the
oracle with the same `dtype`: int8 codes, fp16 scales and sums, and the T-typed outputs of rms_norm bit for bit; silu_and_mul
within 2 ulp of T (approximate exp / reciprocal, as the fp16 kernel and the reference's --use_fast_math build)."""
import numpy as np
import pytest
import torch

from oracle import elementwise as oe
from tests.util import assert_f16_equal, dev

pytestmark = pytest.mark.gpu

DT = {"bf16": torch.bfloat16, "f32": torch.float32}


def _x(tokens, hidden, seed, scale, dtype):
    """-> (oracle-side array, device tensor) of element type `dtype`"""
    f = (np.random.default_rng(seed).standard_normal((tokens, hidden)) * scale).astype(np.float32)
    if dtype == "bf16":
        bits = oe._round_bf16_bits(f)
        return bits, torch.from_numpy(bits.view(np.int16)).to(dev()).view(torch.bfloat16)
    return f, torch.from_numpy(f).to(dev())


def _bits(t):
    return t.cpu().view(torch.int16).numpy().view(np.uint16) if t.dtype == torch.bfloat16 else t.cpu().numpy()


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("tokens,hidden", [(1, 128), (16, 4096), (5, 14336), (3, 28672), (0, 4096)])
@pytest.mark.parametrize("fuse", [False, True])
def test_quant_dtypes(tokens, hidden, fuse, dtype):
    import omniserve_backend.fused_kernels as fk
    x, xd = _x(tokens, hidden, tokens + hidden, 3.0, dtype)
    out = torch.empty((tokens, hidden), dtype=torch.int8, device=dev())
    scale = torch.empty((tokens,), dtype=torch.float16, device=dev())
    ssum = torch.empty((tokens,), dtype=torch.float16, device=dev())
    if fuse:
        fk.invoke_quant_fuse_sum(out, xd, ssum, scale)
    else:
        fk.invoke_quant(out, xd, scale)
    torch.cuda.synchronize()
    if tokens == 0:
        return
    q, s, sm = oe.quant_per_token(x, fuse, dtype=dtype)
    assert np.array_equal(out.cpu().numpy(), q)
    assert_f16_equal(scale, s, "scale")
    if fuse:
        assert_f16_equal(ssum, sm, "sum")


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("tokens,hidden", [(1, 128), (16, 4096), (7, 5120), (4, 96), (33, 8192)])
@pytest.mark.parametrize("fuse", [False, True])
def test_rms_norm_general_dtypes(tokens, hidden, fuse, dtype):
    import omniserve_backend.layernorm_ops as ln
    x, xd = _x(tokens, hidden, 3 * tokens + hidden, 2.0, dtype)
    g, gd = _x(1, hidden, 1, 0.1, dtype)
    g, gd = g[0], gd[0].contiguous()
    out = torch.empty((tokens, hidden), dtype=torch.int8, device=dev())
    scale = torch.empty((tokens,), dtype=torch.float16, device=dev())
    ssum = torch.empty((tokens,), dtype=torch.float16, device=dev())
    if fuse:
        ln.rms_norm_general_fuse_sum(out, xd, gd, ssum, scale, 1e-5, True)
    else:
        ln.rms_norm_general(out, xd, gd, scale, 1e-5, True)
    torch.cuda.synchronize()
    q, s, sm = oe.rms_norm_general(x, g, 1e-5, fuse, dtype=dtype)
    assert np.array_equal(out.cpu().numpy(), q)
    assert_f16_equal(scale, s, "scale")
    if fuse:
        assert_f16_equal(ssum, sm, "sum")


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("tokens,hidden", [(2, 256), (16, 4096), (3, 8192)])
def test_rms_norm_dtypes(tokens, hidden, dtype):
    import omniserve_backend.layernorm_ops as ln
    x, xd = _x(tokens, hidden, 5 * tokens + hidden, 2.0, dtype)
    w, wd = _x(1, hidden, 2, 1.0, dtype)
    w, wd = w[0], wd[0].contiguous()
    out = torch.empty_like(xd)
    ln.rms_norm(out, xd, wd, 1e-5)
    torch.cuda.synchronize()
    want = oe.rms_norm(x, w, 1e-5, dtype=dtype)
    assert np.array_equal(_bits(out), want), "rms_norm(%s) differs from the oracle" % dtype


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("tokens,d", [(1, 64), (16, 14336), (9, 100)])
def test_silu_and_mul_dtypes(tokens, d, dtype):
    import omniserve_backend.activation_ops as act
    x, xd = _x(tokens, 2 * d, 7 * tokens + d, 2.0, dtype)
    out = torch.empty((tokens, d), dtype=DT[dtype], device=dev())
    act.silu_and_mul(out, xd)
    torch.cuda.synchronize()
    want = oe._load(oe.silu_and_mul(x, dtype=dtype), dtype)
    got = out.float().cpu().numpy()
    # bf16 (8 significand bits): <= 2 ulp with v_exp / v_rcp, as the fp16 kernel; fp32 (24 bits): library exp + IEEE
    # division, <= 4 ulp of the oracle's float32 evaluation (two roundings each side)
    ulp = np.abs(want) * (2.0 ** -7 if dtype == "bf16" else 2.0 ** -23) + 1e-30
    err = np.abs(got - want) / ulp
    assert err.max() <= (2.0 if dtype == "bf16" else 4.0), "silu_and_mul(%s): %.2f ulp" % (dtype, err.max())


def test_unsupported_dtype_raises():
    import omniserve_backend.fused_kernels as fk
    x = torch.zeros((2, 128), dtype=torch.float64, device=dev())
    out = torch.empty((2, 128), dtype=torch.int8, device=dev())
    scale = torch.empty((2,), dtype=torch.float16, device=dev())
    with pytest.raises(RuntimeError, match="not implemented for"):
        fk.invoke_quant(out, x, scale)
