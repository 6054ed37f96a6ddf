from __future__ import annotations

import torch

from .. import _lib


def check_gemm_io(in_feats, kernel, out_feats, packed: bool):
    _lib.require_cuda(in_feats, kernel, out_feats)
    if in_feats.dtype != torch.int8 or kernel.dtype != torch.int8 or out_feats.dtype != torch.float16:
        raise RuntimeError("gemm_forward: expected int8 activations/weights and fp16 output")
    if in_feats.dim() != 2 or not in_feats.is_contiguous() or not kernel.is_contiguous():
        raise RuntimeError("gemm_forward: in_feats [M,K] and kernel must be contiguous")
    M, K = in_feats.shape
    N = out_feats.shape[-1]
    if out_feats.shape[-2] != M:
        raise RuntimeError("gemm_forward: out_feats rows != in_feats rows")
    if kernel.shape[0] != N or kernel.shape[1] != (K // 2 if packed else K):
        raise RuntimeError("gemm_forward: weight shape %s does not match N=%d K=%d" % (tuple(kernel.shape), N, K))
    if out_feats.stride(-1) != 1:
        raise RuntimeError("gemm_forward: out_feats rows must be contiguous")
    return M, N, K, out_feats.stride(-2)


def gemm_workspace(M, N, K, device):
    nbytes = _lib.lib().omni_gemm_workspace_bytes(M, N, K)
    ws = _lib.workspace(max(nbytes, 1), device, "gemm")
    return ws
