"""Tensor-parallel sharding of the QServe W4A8 path (BASELINE.json configs[4]: Llama-2-70B TP=8).

The reference has no TP implementation (tp_size is hard-coded to 1,
llama_w4a8_unpad.py:158,529,607-608); this module is the MI355X-native design:

  * one process per GPU (torchrun), `torch.distributed` backend "nccl" = RCCL over xGMI;
  * Megatron split: qkv_proj / gate_up_proj column-parallel (shard the output channels N),
    o_proj / down_proj row-parallel (shard the reduction dim K), attention by KV head;
  * the only collective: an in-place fp16 sum all-reduce of the [M, hidden] projection output
    after o_proj and after down_proj (2 per layer).  The per-channel zero-point term distributes
    over K shards (each rank subtracts szero[n] * sum_{k in shard} x[m,k]) and the per-token
    activation scale may be rank-local, so row-parallel needs no other exchange.

Shard granularity: N in multiples of 64 channels (the kernels' wave tile; covers the 32-row pack
blocks and the 32-channel permutation blocks of the g128 parameters), K in whole 32-k tiles
(multiples of 128 so that g128 groups stay whole).  A K shard is NOT a column slice of the
[N, K/2] view: it is a slice of the [N/32][K/32][512 B] tile view.
"""
from __future__ import annotations

import torch


def shard_range(total: int, rank: int, world: int, align: int):
    if total % (world * align) != 0:
        raise ValueError("dimension %d cannot be split %d-ways in multiples of %d" % (total, world, align))
    per = total // world
    return rank * per, (rank + 1) * per


def shard_qweight_n(qweight: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Column-parallel shard of packed weights [N, K/2]: a row slice (N/world multiple of 64)."""
    n0, n1 = shard_range(qweight.shape[0], rank, world, 64)
    return qweight[n0:n1].contiguous()


def shard_qweight_k(qweight: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Row-parallel shard of packed weights: slice the k-tile axis of the [N/32][K/32][512] view."""
    N, K2 = qweight.shape
    K = 2 * K2
    k0, k1 = shard_range(K, rank, world, 128)
    tiles = qweight.reshape(N // 32, K // 32, 512)
    return tiles[:, k0 // 32:k1 // 32].contiguous().reshape(N, (k1 - k0) // 2)


def shard_vec_n(v: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    n0, n1 = shard_range(v.shape[-1], rank, world, 64)
    return v[..., n0:n1].contiguous()


def shard_group_params_k(p: torch.Tensor, rank: int, world: int, group: int = 128) -> torch.Tensor:
    """g128 second-level params [K/G, N] for a row-parallel (K) shard: a row slice."""
    g0, g1 = shard_range(p.shape[0] * group, rank, world, group)
    return p[g0 // group:g1 // group].contiguous()


def all_reduce_(buf: torch.Tensor, group=None) -> torch.Tensor:
    """In-place fp16 sum all-reduce of a projection output (RCCL on GPU, gloo in the CPU tests)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if buf.is_cuda and dist.get_backend(group) == "gloo":
            # test rigs only (several ranks on one GPU): gloo reduces on the host.  fp32 staging: the sum of
            # `world` fp16 values rounded once, as a 2-rank fp16 ring all-reduce would produce.
            host = buf.float().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            buf.copy_(host.to(buf.dtype))
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf
