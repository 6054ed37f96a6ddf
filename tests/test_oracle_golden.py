"""Pin the oracle against fixtures produced by the reference's own Python packer
(tests/golden/make_golden.py) and against its own invariants."""
import os

import numpy as np

from oracle import w4a8


def test_pack_per_channel_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "w4a8_pack_per_chn.npz"))
    qw, s1, sz = w4a8.pack_per_channel(g["codes"], g["zeros"], g["s1"])
    assert np.array_equal(qw, g["qweight"])
    assert np.array_equal(s1.view(np.uint16), g["s1_scales"].view(np.uint16))
    assert np.array_equal(sz.view(np.uint16), g["s1_szeros"].view(np.uint16))
    N, K = g["codes"].shape
    assert np.array_equal(w4a8.unpack_w4(g["qweight"], N, K), g["codes"])


def test_pack_per_group_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "w4a8_pack_per_group.npz"))
    qw, s1, s2s, s2z = w4a8.pack_per_group(g["codes"], g["zeros"], g["s2"], g["s1"])
    assert np.array_equal(qw, g["qweight"])
    assert np.array_equal(s2s, g["s2_scales"])
    assert np.array_equal(s2z, g["s2_zeros"])
    N, K = g["codes"].shape
    # second-level dequant reproduces the level-1 int8 weights (u - z) * s2
    w8 = w4a8.dequant_per_group_w8(g["qweight"], g["s2_scales"], g["s2_zeros"], N, K)
    ref = (g["codes"].astype(np.int64).reshape(N, K // 128, 128) - g["zeros"][:, :, None]) * g["s2"][:, :, None]
    assert np.array_equal(w8.astype(np.int64), ref.reshape(N, K))


def test_pack_roundtrip_random():
    rng = np.random.default_rng(0)
    u = rng.integers(0, 16, size=(96, 160), dtype=np.uint8)
    assert np.array_equal(w4a8.unpack_w4(w4a8.pack_w4(u), 96, 160), u)
    p = rng.integers(-8, 8, size=(96, 5))
    assert np.array_equal(w4a8.unpermute_group_param(w4a8.permute_group_param(p)), p)


def test_gemm_per_channel_is_affine_dequant():
    """out ~= A_deq @ W_deq^T with W_deq = (u - z)*s1 when asum = sum of dequantized A."""
    N, K, M = 64, 128, 8
    u, z, s1 = w4a8.synth_per_channel(N, K, seed=3)
    qw, s1h, szh = w4a8.pack_per_channel(u, z, s1)
    rng = np.random.default_rng(4)
    a = rng.integers(-127, 128, size=(M, K), dtype=np.int8)
    sa = rng.uniform(0.01, 0.05, size=(M,)).astype(np.float16)
    asum = (a.astype(np.float64).sum(1) * sa.astype(np.float64)).astype(np.float16)
    out = w4a8.gemm_per_chn(a, qw, s1h, sa, szh, asum).astype(np.float64)
    wd = (u.astype(np.float64) - z[:, None]) * s1.astype(np.float64)[:, None]
    ad = a.astype(np.float64) * sa.astype(np.float64)[:, None]
    ref = ad @ wd.T
    assert np.allclose(out, ref, rtol=2e-2, atol=0.3)


def test_gemm_per_group_wrap_semantics():
    """The adversarial set overflows bytes; the oracle must follow the 32-bit
    multiply + per-byte add, not the idealised (u - z)*s2."""
    N, K = 32, 128
    u, z, s2, s1 = w4a8.synth_per_group(N, K, seed=5, wrap=True)
    qw, s1h, s2s, s2z = w4a8.pack_per_group(u, z, s2, s1)
    w8 = w4a8.dequant_per_group_w8(qw, s2s, s2z, N, K).astype(np.int64)
    ideal = ((u.astype(np.int64).reshape(N, 1, K) - z[:, :, None]) * s2[:, :, None]).reshape(N, K)
    assert (w8 != ideal).any()                      # wrap really happened
    ok = np.abs(ideal) <= 127
    # where no byte of the word overflowed the result is the idealised one
    word_ok = (u.astype(np.int64) * np.repeat(s2, K, axis=1) <= 255).reshape(N, K // 4, 4).all(-1)
    mask = np.repeat(word_ok, 4, axis=1) & ok
    assert np.array_equal(w8[mask], ideal[mask])
