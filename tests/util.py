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
