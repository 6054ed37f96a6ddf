"""world_size-2 gloo tests (CPU) of the multi-GPU paths: TP sharding rules of omniserve_amd/tp.py
checked through the oracle GEMM, and the max-over-ranks timing reduction bench.py uses."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import elementwise as oe
from oracle import w4a8
from omniserve_amd import tp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _case():
    N, K, M = 256, 512, 8
    u, z, s1 = w4a8.synth_per_channel(N, K, seed=7)
    qw, s1h, szh = w4a8.pack_per_channel(u, z, s1)
    x = np.random.default_rng(8).standard_normal((M, K)).astype(np.float16)
    return N, K, M, u, z, s1, qw, s1h, szh, x


def _column_parallel(rank, world):
    N, K, M, u, z, s1, qw, s1h, szh, x = _case()
    q, sa, asum = oe.quant_per_token(x, True)
    full = w4a8.gemm_per_chn(q, qw, s1h, sa, szh, asum)
    qw_r = tp.shard_qweight_n(torch.from_numpy(qw), rank, world).numpy()
    s1_r = tp.shard_vec_n(torch.from_numpy(s1h), rank, world).numpy()
    sz_r = tp.shard_vec_n(torch.from_numpy(szh), rank, world).numpy()
    part = w4a8.gemm_per_chn(q, qw_r, s1_r, sa, sz_r, asum)
    gathered = [torch.empty(part.shape, dtype=torch.float16) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(part))
    got = torch.cat(gathered, dim=1).numpy()
    return bool(np.array_equal(got.view(np.uint16), full.view(np.uint16)))


def _row_parallel(rank, world):
    N, K, M, u, z, s1, qw, s1h, szh, x = _case()
    k0, k1 = tp.shard_range(K, rank, world, 128)
    # each rank quantises its own K shard of the activations (scale and sum are rank-local)
    q_r, sa_r, asum_r = oe.quant_per_token(x[:, k0:k1], True)
    qw_r = tp.shard_qweight_k(torch.from_numpy(qw), rank, world).numpy()
    assert np.array_equal(w4a8.unpack_w4(qw_r, N, k1 - k0), u[:, k0:k1])     # tile-view slice is right
    part = torch.from_numpy(w4a8.gemm_per_chn(q_r, qw_r, s1h, sa_r, szh, asum_r))
    tp.all_reduce_(part)                                                     # fp16 sum over ranks
    wd = (u.astype(np.float64) - z[:, None]) * s1.astype(np.float64)[:, None]
    ref = x.astype(np.float64) @ wd.T
    err = np.abs(part.numpy().astype(np.float64) - ref)
    return bool((err <= 0.02 * np.abs(ref) + 0.5).all())


def _max_time(rank, world):
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    return float(t.item())


def test_column_parallel_shards_concatenate_bit_exact():
    assert all(_run(_column_parallel))


def test_row_parallel_all_reduce_matches_dequantised_reference():
    assert all(_run(_row_parallel))


def test_bench_time_reduction_is_max_over_ranks():
    assert _run(_max_time) == [2.0, 2.0]


def test_shard_ranges_reject_misaligned_splits():
    import pytest
    with pytest.raises(ValueError):
        tp.shard_range(4096 + 64, 0, 8, 64)
    assert tp.shard_range(8192, 3, 8, 128) == (3072, 4096)
