"""Shared helpers for the parity tests (HIP path vs oracle on identical seeded inputs)."""
import numpy as np
import torch


def dev():
    return torch.device("cuda:0")


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def f16_bits(t):
    if torch.is_tensor(t):
        t = t.detach().cpu().numpy()
    return np.ascontiguousarray(t).view(np.uint16)


def assert_f16_equal(got, want, what=""):
    g, w = f16_bits(got), f16_bits(want)
    bad = g != w
    if bad.any():
        idx = np.argwhere(bad)[:5]
        raise AssertionError("%s: %d / %d fp16 values differ, first at %s: got %s want %s" % (
            what, bad.sum(), bad.size, idx.tolist(),
            np.asarray(got.detach().cpu().numpy() if torch.is_tensor(got) else got)[tuple(idx[0])],
            np.asarray(want)[tuple(idx[0])]))


def f16_ulp_diff(got, want):
    """max |difference| in units of fp16 ulps (monotone integer mapping of the bit patterns)."""
    def key(x):
        b = f16_bits(x).astype(np.int32)
        return np.where(b & 0x8000, -(b & 0x7FFF), b & 0x7FFF)
    return int(np.abs(key(got) - key(want)).max(initial=0))


def attention_errors(got, ref):
    """(max over output vectors of ||got - ref||_2 / ||ref||_2,  max over elements of |got - ref| / max_d |ref[..., :]|):
    north_star's 1e-3 RELATIVE bar taken per output vector (one (token / sequence, head) row of head_dim values) -- no
    tensor-wide absolute term."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    num = np.sqrt(((got - ref) ** 2).sum(axis=-1))
    den = np.sqrt((ref ** 2).sum(axis=-1)) + 1e-30
    head_max = np.abs(ref).max(axis=-1, keepdims=True) + 1e-30
    return float((num / den).max()), float((np.abs(got - ref) / head_max).max())


def assert_attention_close(got, ref, what="attention", bar=1e-3):
    """Per-head bar of tests/test_edge_cases_gpu.py::test_decode_attention_baseline_shapes, with the measured distances in
    the message: for every output vector ||got - ref||_2 <= bar * ||ref||_2 and max |got - ref| <= bar * max |ref|."""
    got = np.asarray(got, np.float32)
    ref = np.asarray(ref, np.float32)
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (what, got.shape, ref.shape)
    assert np.isfinite(got).all(), "%s: non-finite output" % what
    l2, linf = attention_errors(got, ref)
    msg = "%s: per-head rel L2 %.3g, max |d| / head max %.3g (bar %.1e)" % (what, l2, linf, bar)
    print(msg)
    assert l2 <= bar and linf <= bar, msg
    return msg


def quantize_act(x_f16):
    """Oracle-side per-token activation quantisation used to make GEMM inputs."""
    from oracle import elementwise as oe
    q, s, sm = oe.quant_per_token(x_f16, fuse_sum=True)
    return q, s, sm


class GpuPagedKV:
    """GPU mirror of oracle.kv4.PagedKV4: one uint8 pool per K/V, pointer tables."""

    def __init__(self, oracle_k, oracle_v, k_table_idx, v_table_idx):
        self.kpool = to_dev(oracle_k.pool.copy())
        self.vpool = to_dev(oracle_v.pool.copy())
        B, M = k_table_idx.shape
        tab = np.zeros((B, 2, M), np.int64)
        tab[:, 0, :] = self.kpool.data_ptr() + k_table_idx.astype(np.int64) * oracle_k.page_bytes
        tab[:, 1, :] = self.vpool.data_ptr() + v_table_idx.astype(np.int64) * oracle_v.page_bytes
        self.table = to_dev(tab)

    def pools(self):
        return self.kpool.cpu().numpy(), self.vpool.cpu().numpy()
