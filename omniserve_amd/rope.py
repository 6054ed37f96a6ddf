"""Host-built RoPE coefficient tables for the KV4 kernels.

table[pos, i] = (cos, sin)(pos * scale / base**(2i/dim)) in float32, i < dim/2 -- the angle the
reference evaluates on the device (common/decoderMaskedMultiheadAttentionUtils.h:1147-1152).
Building it once on the host (numpy float32) makes the RoPE bit-reproducible and removes
powf/sincosf from the bandwidth-bound kernels.
"""
from __future__ import annotations

import numpy as np
import torch

_cache = {}
_retired = []     # superseded tables stay allocated: captured HIP graphs hold their raw pointers


def build_table_numpy(max_pos: int, dim: int, base: float, scale: float = 1.0) -> np.ndarray:
    half = dim // 2
    i = np.arange(half, dtype=np.float32)
    denom = np.power(np.float32(base), (np.float32(2.0) * i / np.float32(dim)).astype(np.float32)).astype(np.float32)
    t = (np.arange(max_pos, dtype=np.float32)[:, None] * np.float32(scale)).astype(np.float32)
    ang = (t / denom[None, :]).astype(np.float32)
    out = np.empty((max_pos, half, 2), np.float32)
    out[..., 0] = np.cos(ang)
    out[..., 1] = np.sin(ang)
    return out


def rope_table(max_pos: int, dim: int, base: float, scale: float, device) -> torch.Tensor:
    """Cached device table with at least `max_pos` rows (rounded up to a power of two >= 4096)."""
    key = (device, dim, base, scale)
    t = _cache.get(key)
    if t is None or t.shape[0] < max_pos:
        dim, base, scale = int(dim), float(base), float(scale)
        rows = 4096
        while rows < max_pos:
            rows *= 2
        if t is not None:
            _retired.append(t)
        t = torch.from_numpy(build_table_numpy(rows, dim, base, scale)).to(device)
        _cache[key] = t
    return t
