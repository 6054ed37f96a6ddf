"""Oracle (CPU, numpy) for the QServe W4A8 / W8A8 GEMMs.  Test infrastructure only.

Restates, with citations relative to the upstream checkout:

* weight packing       omniserve/modeling/layers/quantized_linear/w4a8_linear.py:141-337
* per-channel GEMM     kernels/csrc/qgemm/w4a8_per_chn/gemm_cuda.cu:281-306,569-598
* per-group  GEMM      kernels/csrc/qgemm/w4a8_per_group/gemm_cuda.cu:276-331,616-629
* W8A8 GEMM            kernels/csrc/qgemm/w8a8/w8a8_gemm_cuda.cu (epilogue as per-group)

Integer accumulation is exact (int64 -> int32 range checked); the epilogues
are evaluated in float32 in the reference's operand order without FMA
contraction (the CUDA build is --use_fast_math, so its own contraction is
compiler-chosen; the HIP kernels compile their epilogues with
-ffp-contract=off and are bit-identical to this file).
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# packing
# --------------------------------------------------------------------------

def pack_w4(u: np.ndarray) -> np.ndarray:
    """uint4 codes U[N,K] (values 0..15) -> packed int8 [N, K/2].

    Closed form of the two permute/contiguous steps at w4a8_linear.py:296-327:
    tile (nb,kb) of 32x32 codes occupies 512 contiguous bytes
    ``[lane = n3*4 + k6][byte = k5*8 + n2*4 + k7]`` and nibble n1 (0 = low)
    with n = nb*32 + n1*16 + n2*8 + n3 and k = kb*32 + k5*16 + k6*4 + k7.
    """
    u = np.asarray(u)
    N, K = u.shape
    assert N % 32 == 0 and K % 32 == 0
    assert u.min() >= 0 and u.max() <= 15
    t = u.astype(np.uint8).reshape(N // 32, 2, 2, 8, K // 32, 2, 4, 4)
    #           axes:            nb   n1 n2 n3   kb     k5 k6 k7
    t = t.transpose(0, 4, 3, 6, 5, 2, 7, 1)  # nb kb n3 k6 k5 n2 k7 n1
    packed = (t[..., 1] << 4) | t[..., 0]
    return np.ascontiguousarray(packed).reshape(N, K // 2).view(np.int8)


def unpack_w4(q: np.ndarray, N: int, K: int) -> np.ndarray:
    """Inverse of :func:`pack_w4` -> uint8 codes [N,K]."""
    b = np.asarray(q).view(np.uint8).reshape(N // 32, K // 32, 8, 4, 2, 2, 4)
    lo = b & 0xF
    hi = b >> 4
    t = np.stack([lo, hi], axis=-1)  # nb kb n3 k6 k5 n2 k7 n1
    t = t.transpose(0, 7, 5, 2, 1, 4, 3, 6)  # nb n1 n2 n3 kb k5 k6 k7
    return np.ascontiguousarray(t).reshape(N, K)


def permute_group_param(p: np.ndarray) -> np.ndarray:
    """[N, K/G] per-(channel,group) parameter -> stored [K/G, N] with the
    per-32-channel permutation ``pos = i*4 + j  <->  n = j*8 + i``
    (w4a8_linear.py:236-282).
    """
    N, NG = p.shape
    t = p.T.reshape(NG, N // 32, 4, 8).transpose(0, 1, 3, 2)
    return np.ascontiguousarray(t).reshape(NG, N)


def unpermute_group_param(s: np.ndarray) -> np.ndarray:
    NG, N = s.shape
    t = s.reshape(NG, N // 32, 8, 4).transpose(0, 1, 3, 2)
    return np.ascontiguousarray(t).reshape(NG, N).T


def pack_per_channel(u, zeros, s1):
    """(codes U[N,K], integer zero points[N], fp16 scales s1[N]) ->
    (qweight int8[N,K/2], s1_scales fp16[N], s1_szeros fp16[N]).
    s1_szeros = zeros * s1 evaluated in fp16 (w4a8_linear.py:333-335)."""
    s1 = np.asarray(s1, dtype=np.float16)
    z = np.asarray(zeros).astype(np.float16)
    return pack_w4(u), s1.copy(), (z * s1).astype(np.float16)


def pack_per_group(u, zeros, s2, s1):
    """(codes U[N,K], zero points [N,K/G] in 0..15, int scales s2[N,K/G],
    fp16 s1[N]) -> (qweight, s1_scales, s2_scales int8[K/G,N], s2_zeros int8[K/G,N]).
    s2_zeros = int8((-zero) * s2) after permutation (w4a8_linear.py:257-282)."""
    s2p = permute_group_param(np.asarray(s2).astype(np.int64))
    zp = permute_group_param(-np.asarray(zeros).astype(np.int64))
    z2 = (zp * s2p)
    return (pack_w4(u), np.asarray(s1, dtype=np.float16).copy(),
            s2p.astype(np.int8), z2.astype(np.int8))


# --------------------------------------------------------------------------
# GEMMs
# --------------------------------------------------------------------------

def _int_matmul(a_i8: np.ndarray, w_i8: np.ndarray) -> np.ndarray:
    """acc[m,n] = sum_k a[m,k]*w[n,k], exact, returned as int32."""
    acc = a_i8.astype(np.int64) @ w_i8.astype(np.int64).T
    assert np.abs(acc).max(initial=0) < 2**31
    return acc.astype(np.int32)


def _int_matmul_fast(a_i8, w_i8):
    """Same result through float64 BLAS (exact while |acc| < 2**53)."""
    acc = a_i8.astype(np.float64) @ w_i8.astype(np.float64).T
    return np.rint(acc).astype(np.int64).astype(np.int32)


def gemm_per_chn_acc(a_i8, qweight):
    M, K = a_i8.shape
    N = qweight.shape[0]
    u = unpack_w4(qweight, N, K)
    return _int_matmul_fast(a_i8, u)


def gemm_per_chn(a_i8, qweight, wscales, ascales, w_szs, a_ssums):
    """out = h( f32(acc)*s_w[n]*s_a[m] - sz[n]*asum[m] ), acc = sum_k A*U, U in 0..15.
    (w4a8_per_chn/gemm_cuda.cu:585-593)."""
    acc = gemm_per_chn_acc(a_i8, qweight).astype(np.float32)
    sw = np.asarray(wscales, np.float16).astype(np.float32)[None, :]
    sa = np.asarray(ascales, np.float16).astype(np.float32)[:, None]
    sz = np.asarray(w_szs, np.float16).astype(np.float32)[None, :]
    asum = np.asarray(a_ssums, np.float16).astype(np.float32)[:, None]
    t = (acc * sw).astype(np.float32)
    t = (t * sa).astype(np.float32)
    c = (sz * asum).astype(np.float32)
    return (t - c).astype(np.float32).astype(np.float16)


def dequant_per_group_w8(qweight, s2_scales, s2_zeros, N, K, G=128):
    """Second-level dequant to int8, byte arithmetic exactly as the reference
    (w4a8_per_group/gemm_cuda.cu:286-329): four codes of one channel that are
    adjacent in k sit in one 32-bit word ``0x0d0c0b0a``; the word is multiplied
    by the (unsigned) scale byte as a 32-bit integer (so a byte product above
    255 carries into its neighbour) and the zero byte is added per byte mod 256
    (``__vadd4``); each resulting byte is reinterpreted as int8."""
    u = unpack_w4(qweight, N, K).astype(np.uint32)       # [N,K]
    s2 = unpermute_group_param(np.asarray(s2_scales).view(np.uint8)).astype(np.uint32)  # [N,K/G]
    z2 = unpermute_group_param(np.asarray(s2_zeros).view(np.uint8)).astype(np.uint32)
    w = u.reshape(N, K // 4, 4)
    word = (w[..., 0] | (w[..., 1] << 8) | (w[..., 2] << 16) | (w[..., 3] << 24)).astype(np.uint64)
    s = np.repeat(s2, G // 4, axis=1).astype(np.uint64)   # [N, K/4]
    z = np.repeat(z2, G // 4, axis=1).astype(np.uint32)
    prod = ((word * s) & 0xFFFFFFFF).astype(np.uint32)
    out = np.empty((N, K // 4, 4), np.uint8)
    for b in range(4):
        out[..., b] = (((prod >> (8 * b)) & 0xFF) + z) & 0xFF
    return out.reshape(N, K).view(np.int8)


def gemm_per_group_acc(a_i8, qweight, s2_zeros, s2_scales, G=128):
    M, K = a_i8.shape
    N = qweight.shape[0]
    w8 = dequant_per_group_w8(qweight, s2_scales, s2_zeros, N, K, G)
    return _int_matmul_fast(a_i8, w8)


def gemm_per_group(a_i8, qweight, s2_zeros, s2_scales, wscales, ascales, G=128):
    """out = h( f32(acc) * (s1[n]*s_a[m]) ) (w4a8_per_group/gemm_cuda.cu:620-627)."""
    acc = gemm_per_group_acc(a_i8, qweight, s2_zeros, s2_scales, G).astype(np.float32)
    sw = np.asarray(wscales, np.float16).astype(np.float32)[None, :]
    sa = np.asarray(ascales, np.float16).astype(np.float32)[:, None]
    s = (sw * sa).astype(np.float32)
    return (acc * s).astype(np.float32).astype(np.float16)


def gemm_w8a8(a_i8, w_i8, wscales, ascales):
    """out = h( f32(sum A*W8) * (s_w[n]*s_a[m]) ); W int8 [N,K] row-major
    (w8a8/w8a8_gemm_cuda.cu epilogue, same form as per-group)."""
    acc = _int_matmul_fast(a_i8, w_i8).astype(np.float32)
    sw = np.asarray(wscales, np.float16).astype(np.float32)[None, :]
    sa = np.asarray(ascales, np.float16).astype(np.float32)[:, None]
    s = (sw * sa).astype(np.float32)
    return (acc * s).astype(np.float32).astype(np.float16)


# --------------------------------------------------------------------------
# deterministic synthetic inputs
# --------------------------------------------------------------------------

def synth_per_channel(N, K, seed=0):
    rng = np.random.default_rng(seed)
    u = rng.integers(0, 16, size=(N, K), dtype=np.uint8)
    zeros = rng.integers(0, 16, size=(N,), dtype=np.int64)
    s1 = rng.uniform(0.002, 0.02, size=(N,)).astype(np.float16)
    return u, zeros, s1


def synth_per_group(N, K, G=128, seed=0, wrap=False):
    """``wrap=False`` keeps the checkpoint invariant |(u - z)*s2| <= 127
    (s2 in 1..8); ``wrap=True`` is the adversarial set whose byte arithmetic
    overflows (s2 up to 40: byte products above 255 carry)."""
    rng = np.random.default_rng(seed)
    u = rng.integers(0, 16, size=(N, K), dtype=np.uint8)
    zeros = rng.integers(0, 16, size=(N, K // G), dtype=np.int64)
    hi = 41 if wrap else 9
    s2 = rng.integers(1, hi, size=(N, K // G), dtype=np.int64)
    s1 = rng.uniform(0.002, 0.02, size=(N,)).astype(np.float16)
    return u, zeros, s2, s1
