"""Parity of the off-path overloads (csrc/offpath.hip) against oracle/elementwise.py, called through the mirror modules
with the reference's positional signatures (kernels/csrc/fused.cpp:16-76, layernorm.cpp:17-77, activation.cpp).

Integer-exact stages (static quantisers, dequantisers, the norms' codes and residual update) are compared bit for bit;
where the reference itself is --use_fast_math (tanh, exp, 1/x) the tolerance is 1 fp16 ulp / 1 int8 code and written
next to the check."""
import numpy as np
import pytest
import torch

from oracle import elementwise as oe
from tests.util import assert_f16_equal, dev, f16_ulp_diff, to_dev

pytestmark = pytest.mark.gpu


def _x(tokens, hidden, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal((tokens, hidden)) * scale).astype(np.float16)


def _acc(tokens, cols, seed, span=60000):
    return np.random.default_rng(seed).integers(-span, span, (tokens, cols)).astype(np.int32)


@pytest.mark.parametrize("tokens,hidden", [(1, 8), (16, 4096), (5, 14336), (1100, 4096), (0, 4096)])
@pytest.mark.parametrize("fuse", [False, True])
def test_quant_static(tokens, hidden, fuse):
    import omniserve_backend.fused_kernels as fk
    x = _x(tokens, hidden, tokens + hidden, 3.0)
    out = torch.full((tokens, hidden), 99, dtype=torch.int8, device=dev())
    if fuse:
        fk.invoke_quant_fuse_sum(out, to_dev(x), 0.0, 0.0437)     # the scalar input_sum is ignored upstream too
    else:
        fk.invoke_quant(out, to_dev(x), 0.0437)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), oe.quant_static(x, 0.0437))      # bit-exact (IEEE division both sides)


def test_quant_static_saturates_and_rounds_half_to_even():
    import omniserve_backend.fused_kernels as fk
    x = np.array([[0.25, 0.75, 1.25, -0.25, -0.75, 100.0, -100.0, 0.0]], np.float16)
    out = torch.empty((1, 8), dtype=torch.int8, device=dev())
    fk.invoke_quant(out, to_dev(x), 0.5)
    assert out.cpu().numpy().tolist() == [[0, 2, 2, 0, -2, 127, -128, 0]]


@pytest.mark.parametrize("tokens,hidden,pad", [(3, 64, 0), (16, 4096, 0), (9, 1024, 64), (1100, 4096, 8)])
def test_dequant_strided_rows(tokens, hidden, pad):
    import omniserve_backend.fused_kernels as fk
    acc = _acc(tokens, hidden + pad, 5)
    full_in = to_dev(acc)
    full_out = torch.zeros((tokens, hidden + 2 * pad), dtype=torch.float16, device=dev())
    fk.invoke_dequant(full_out[:, :hidden], full_in[:, :hidden], 0.00317)
    torch.cuda.synchronize()
    assert_f16_equal(full_out[:, :hidden], oe.dequant(acc[:, :hidden], 0.00317), "dequant")
    assert float(full_out[:, hidden:].abs().max()) == 0.0 if pad else True    # nothing written past the row


@pytest.mark.parametrize("tokens,hidden", [(1, 128), (16, 4096), (1030, 8192)])
@pytest.mark.parametrize("per_token", [False, True])
def test_dequant_add_residual(tokens, hidden, per_token):
    import omniserve_backend.fused_kernels as fk
    acc, res = _acc(tokens, hidden, 7), _x(tokens, hidden, 8, 2.0)
    out = torch.empty((tokens, hidden), dtype=torch.float16, device=dev())
    if per_token:
        sc = (0.0002 + 0.0003 * np.random.default_rng(9).random(tokens)).astype(np.float16)
        fk.invoke_dequant_add_residual(out, to_dev(acc), to_dev(res), to_dev(sc))
    else:
        sc = 0.00031
        fk.invoke_dequant_add_residual(out, to_dev(acc), to_dev(res), sc)
    torch.cuda.synchronize()
    assert_f16_equal(out, oe.dequant_add_residual(acc, res, sc), "dequant_add_residual")     # one FMA, bit-exact


@pytest.mark.parametrize("tokens,hidden", [(1, 128), (16, 4096), (7, 5120), (33, 8192), (1100, 4096), (2, 16256)])
def test_rms_norm_use_quant(tokens, hidden):
    import omniserve_backend.layernorm_ops as ln
    x = _x(tokens, hidden, tokens + 1, 2.0)
    w = (20.0 * (1.0 + 0.1 * np.random.default_rng(2).standard_normal(hidden))).astype(np.float16)
    out = torch.empty((tokens, hidden), dtype=torch.int8, device=dev())
    ln.rms_norm(out, to_dev(x), to_dev(w), 1e-5, True)
    torch.cuda.synchronize()
    q = oe.rms_norm_quant(x, w, 1e-5)
    assert np.abs(q).max() > 40          # the codes actually span the int8 range
    assert np.array_equal(out.cpu().numpy(), q)


@pytest.mark.parametrize("tokens,hidden", [(1, 128), (4, 200), (16, 4096), (7, 5120), (1100, 4096)])
def test_rms_norm_general_per_tensor(tokens, hidden):
    import omniserve_backend.layernorm_ops as ln
    x = _x(tokens, hidden, 3 * tokens + hidden, 2.0) + np.float16(0.25)     # non-zero mean matters
    g = (1.0 + 0.1 * np.random.default_rng(1).standard_normal(hidden)).astype(np.float16)
    scaling = np.array([23.5], np.float16)
    out = torch.empty((tokens, hidden), dtype=torch.int8, device=dev())
    ln.rms_norm_general(out, to_dev(x), to_dev(g), to_dev(scaling), 1e-5, False)
    torch.cuda.synchronize()
    q = oe.rms_norm_general_static(x, g, scaling, 1e-5)
    assert np.abs(q).max() > 40
    assert np.array_equal(out.cpu().numpy(), q)
    with pytest.raises(AssertionError):          # the reference asserts on the per-tensor fuse_sum form
        ln.rms_norm_general_fuse_sum(out, to_dev(x), to_dev(g), to_dev(scaling), to_dev(scaling), 1e-5, False)


@pytest.mark.parametrize("tokens,hidden", [(1, 128), (16, 4096), (5, 8192), (1100, 4096)])
@pytest.mark.parametrize("per_token", [False, True])
def test_dequant_add_residual_rms_norm_quant(tokens, hidden, per_token):
    import omniserve_backend.layernorm_ops as ln
    acc, res = _acc(tokens, hidden, 11), _x(tokens, hidden, 12, 2.0)
    g = (25.0 * (1.0 + 0.1 * np.random.default_rng(3).standard_normal(hidden))).astype(np.float16)
    out = torch.empty((tokens, hidden), dtype=torch.int8, device=dev())
    r = to_dev(res.copy())
    if per_token:
        sc = (0.0002 + 0.0003 * np.random.default_rng(13).random(tokens)).astype(np.float16)
        ln.invoke_dequant_add_residual_rms_norm_quant(out, to_dev(acc), r, to_dev(g), to_dev(sc), 1e-6)
    else:
        sc = 0.00027
        ln.invoke_dequant_add_residual_rms_norm_quant(out, to_dev(acc), r, to_dev(g), sc, 1e-6)
    torch.cuda.synchronize()
    q, new_res = oe.dequant_add_residual_rms_norm_quant(acc, res, g, sc, 1e-6)
    assert_f16_equal(r, new_res, "residual (updated in place)")
    assert np.abs(q).max() > 40
    assert np.array_equal(out.cpu().numpy(), q)       # variance by ordered FMAs + the reference tree: bit-exact


@pytest.mark.parametrize("tokens,d", [(1, 8), (16, 4096), (3, 11008), (1100, 2048)])
@pytest.mark.parametrize("fast", [False, True])
def test_gelu(tokens, d, fast):
    import omniserve_backend.activation_ops as act
    x = _x(tokens, d, tokens + d, 2.5)
    special = np.array([0.0, -0.0, 1e-4, -8.0, 8.0, 65504.0], np.float16)
    x.reshape(-1)[: min(6, x.size)] = special[: min(6, x.size)]
    out = torch.empty((tokens, d), dtype=torch.float16, device=dev())
    (act.gelu_fast if fast else act.gelu_new)(out, to_dev(x))
    torch.cuda.synchronize()
    want = (oe.gelu_fast if fast else oe.gelu_new)(x)
    got = out.cpu().numpy()
    finite = np.isfinite(want.astype(np.float32))
    assert np.array_equal(np.isfinite(got.astype(np.float32)), finite)
    # the kernel's tanh (v_exp / v_rcp, absolute error ~1e-7) vs libm: at most 1 fp16 ulp after the two roundings, rarely
    assert f16_ulp_diff(got[finite], want[finite]) <= 1
    assert (got[finite] == want[finite]).mean() > 0.995
    ref = torch.nn.functional.gelu(torch.from_numpy(x.astype(np.float32)), approximate="tanh").numpy()
    ok = finite & (np.abs(x.astype(np.float32)) < 100)
    assert np.abs(got.astype(np.float32)[ok] - ref[ok]).max() < 6e-3        # and it IS the tanh GELU


@pytest.mark.parametrize("tokens,d", [(1, 8), (16, 14336), (3, 20000), (1100, 1024)])
def test_dequant_silu_and_mul_quant(tokens, d):
    import omniserve_backend.activation_ops as act
    acc = _acc(tokens, 2 * d, 21, 40000)
    sg, su = 1.1e-4, 0.9e-4
    # static output scale
    out = torch.empty((tokens, d), dtype=torch.int8, device=dev())
    act.invoke_dequant_silu_and_mul_quant(out, to_dev(acc), sg, su, 0.05)
    torch.cuda.synchronize()
    q = oe.dequant_silu_and_mul_quant(acc, sg, su, 0.05)
    diff = np.abs(out.cpu().numpy().astype(np.int32) - q.astype(np.int32))
    assert diff.max() <= 1 and (diff != 0).mean() < 2e-3      # v_exp / v_rcp vs libm: a code moves only at a rounding tie
    # per-token scale (+ the f32 scratch the reference's wrapper allocates, activation.py:126-133)
    out2 = torch.empty((tokens, d), dtype=torch.int8, device=dev())
    scale = torch.empty((tokens,), dtype=torch.float32, device=dev())
    tmp = torch.empty((tokens, d), dtype=torch.float32, device=dev())
    act.invoke_dequant_silu_and_mul_quant(out2, to_dev(acc), sg, su, scale, tmp)
    torch.cuda.synchronize()
    q2, s2, t2 = oe.dequant_silu_and_mul_quant(acc, sg, su)
    np.testing.assert_allclose(tmp.cpu().numpy(), t2, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(scale.cpu().numpy(), s2, rtol=2e-6)
    diff = np.abs(out2.cpu().numpy().astype(np.int32) - q2.astype(np.int32))
    assert diff.max() <= 1 and (diff != 0).mean() < 2e-3
    assert np.abs(out2.cpu().numpy()).max() == 127          # amax maps to +-127


def test_shapes_the_kernels_do_not_cover_are_rejected():
    import omniserve_backend.fused_kernels as fk
    import omniserve_backend.layernorm_ops as ln
    x = to_dev(_x(2, 100, 0))
    with pytest.raises(RuntimeError):
        fk.invoke_quant(torch.empty((2, 100), dtype=torch.int8, device=dev()), x, 0.5)      # rows % 8
    big = to_dev(_x(1, 16384, 0))
    with pytest.raises(RuntimeError):
        ln.rms_norm(torch.empty((1, 16384), dtype=torch.int8, device=dev()), big, big[0], 1e-5, True)
