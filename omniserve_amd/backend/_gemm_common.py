"""Shared marshalling of the three GEMM mirrors.  These functions run once per projection of every decode step of an eager
(not graph-captured) caller -- the reference's own stack -- where the host, not the GPU, sets the pace (bench.py `drop_in`):
every tensor attribute below is read once, sizes of the scratch are cached per shape."""
from __future__ import annotations

import torch

from .. import _lib

_I8, _F16 = torch.int8, torch.float16
_ws_bytes = {}          # (M, N, K) -> omni_gemm_workspace_bytes


def check_gemm_io(in_feats, kernel, out_feats, packed: bool):
    if not (in_feats.is_cuda and kernel.is_cuda and out_feats.is_cuda):
        _lib.require_cuda(in_feats, kernel, out_feats)          # raises (or is patched out by the CPU test harness)
    if in_feats.dtype is not _I8 or kernel.dtype is not _I8 or out_feats.dtype is not _F16:
        raise RuntimeError("gemm_forward: expected int8 activations/weights and fp16 output")
    ishape, kshape, oshape = in_feats.shape, kernel.shape, out_feats.shape
    if len(ishape) != 2 or not in_feats.is_contiguous() or not kernel.is_contiguous():
        raise RuntimeError("gemm_forward: in_feats [M,K] and kernel must be contiguous")
    M, K = ishape
    N = oshape[-1]
    if oshape[-2] != M:
        raise RuntimeError("gemm_forward: out_feats rows != in_feats rows")
    if kshape[0] != N or kshape[1] != (K // 2 if packed else K):
        raise RuntimeError("gemm_forward: weight shape %s does not match N=%d K=%d" % (tuple(kshape), N, K))
    ostride = out_feats.stride()
    if ostride[-1] != 1:
        raise RuntimeError("gemm_forward: out_feats rows must be contiguous")
    return M, N, K, ostride[-2]


def gemm_workspace(M, N, K, device):
    key = (M, N, K)
    nbytes = _ws_bytes.get(key)
    if nbytes is None:
        nbytes = _ws_bytes[key] = max(int(_lib.lib().omni_gemm_workspace_bytes(M, N, K)), 1)
    return _lib.workspace(nbytes, device, "gemm")
