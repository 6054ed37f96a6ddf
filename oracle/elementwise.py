"""Oracle (CPU, numpy) for the activation-quant / norm / activation kernels.
Test infrastructure only.

Restates (citations relative to the upstream checkout):

* invoke_quant / invoke_quant_fuse_sum   kernels/csrc/fused_kernels.cu:57-142
* rms_norm_general[_fuse_sum]            kernels/csrc/layernorm_kernels.cu:58-331, launch :432-513
* rms_norm                               kernels/csrc/layernorm_kernels.cu:335-365
* silu_and_mul                           kernels/csrc/activation_kernels.cu:10-30
* block reductions                       kernels/csrc/reduction_utils.cuh:25-164
* float->int8                            kernels/csrc/utils.cuh:79-84  (cvt.rni.sat.s8.f32)

The reference's block reductions are a 32-lane xor butterfly followed by a
butterfly over the (zero padded) 32 warp partials; `ref_tree_sum` reproduces
that exact float32 summation tree, and the HIP kernels use the same tree, so
sums agree bit-for-bit with this file.  Division and rsqrt are IEEE here
(the CUDA build uses --use_fast_math approximations, which are not
reproducible off NVIDIA hardware).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
F16 = np.float16


def rni_sat_s8(x: np.ndarray) -> np.ndarray:
    """cvt.rni.sat.s8.f32: round-half-even, saturate, NaN -> 0 (utils.cuh:79-84)."""
    x = np.asarray(x, dtype=F32)
    r = np.rint(x)
    r = np.where(np.isnan(r), 0.0, r)
    return np.clip(r, -128, 127).astype(np.int8)


def _butterfly32(v: np.ndarray, op) -> np.ndarray:
    """v[..., 32] -> all-lanes result of xor-shuffle butterfly (mask 16,8,4,2,1)."""
    idx = np.arange(32)
    for mask in (16, 8, 4, 2, 1):
        v = op(v, v[..., idx ^ mask])
    return v


def ref_tree_sum(partials: np.ndarray) -> np.ndarray:
    """blockReduceSum / blockAllReduceSum (reduction_utils.cuh:47-85) over the
    last axis (= threads, a multiple of 32, at most 1024), float32."""
    p = np.asarray(partials, dtype=F32)
    nt = p.shape[-1]
    assert nt % 32 == 0 and nt <= 1024
    w = _butterfly32(p.reshape(p.shape[:-1] + (nt // 32, 32)), lambda a, b: (a + b).astype(F32))[..., 0]
    pad = np.zeros(w.shape[:-1] + (32,), F32)
    pad[..., : nt // 32] = w
    return _butterfly32(pad, lambda a, b: (a + b).astype(F32))[..., 0]


def ref_tree_max(partials: np.ndarray, pad_value=-1e20) -> np.ndarray:
    p = np.asarray(partials, dtype=F32)
    return p.max(axis=-1)  # max is order independent


def _thread_partials(x: np.ndarray, nthreads: int, fn, init):
    """Per-thread sequential accumulation over elements t, t+nthreads, ...
    x: [tokens, hidden] float32.  Returns [tokens, nthreads]."""
    tokens, hidden = x.shape
    acc = np.full((tokens, nthreads), init, dtype=F32)
    for start in range(0, hidden, nthreads):
        chunk = x[:, start:start + nthreads]
        n = chunk.shape[1]
        acc[:, :n] = fn(acc[:, :n], chunk)
    return acc


def quant_per_token(x_h: np.ndarray, fuse_sum: bool):
    """invoke_quant / invoke_quant_fuse_sum, tensor-scale overloads.

    x fp16 [tokens, hidden] -> (q int8, scale fp16[tokens], sum fp16[tokens] | None).
    block = min(hidden,1024); amax starts at 0 (no floor: an all-zero row
    divides by zero exactly as the reference does); scale = h(amax/127);
    q = rni_sat(x * (127/amax)); sum = h(tree_sum(per-thread f32 sums)).
    """
    x = np.asarray(x_h, dtype=F16).astype(F32)
    tokens, hidden = x.shape
    nt = min(hidden, 1024)
    amax = np.abs(x).max(axis=1).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = (amax / F32(127.0)).astype(F32).astype(F16)
        tmp = (F32(127.0) / amax).astype(F32)
        q = rni_sat_s8((x * tmp[:, None]).astype(F32))
    s = None
    if fuse_sum:
        ntp = ((nt + 31) // 32) * 32
        part = _thread_partials(x, nt, lambda a, c: (a + c).astype(F32), 0.0)
        if ntp != nt:
            part = np.concatenate([part, np.zeros((tokens, ntp - nt), F32)], axis=1)
        s = ref_tree_sum(part).astype(F16)
    return q, scale, s


def rms_norm_general(x_h, gamma_h, eps: float, fuse_sum: bool):
    """generalLayerNorm[_fuse_sum]<half, at::Half>, per-token path, use_shmem=false.

    y = (x - mean) * rsqrt(mean(x^2) + eps) * gamma in f32  [mean IS subtracted
    in the output but NOT in the variance: layernorm_kernels.cu:26-34,110-113,127];
    yh = h(y); amax = max(1e-6 (as fp16), |yh|) in fp16; per-thread sum
    accumulates yh in fp16 (`T_scalar sum`, :280,291) then the f32 block tree;
    q = rni_sat( y_f32 * (127/amax) ) recomputed from f32 (:308-318);
    scale = h(amax/127).   block = roundup32(min(hidden,1024)).
    """
    x = np.asarray(x_h, dtype=F16).astype(F32)
    g = np.asarray(gamma_h, dtype=F16).astype(F32)
    tokens, hidden = x.shape
    nt = ((min(hidden, 1024) + 31) // 32) * 32
    eps = F32(eps)

    def pad(part):
        return part

    psum = _thread_partials(x, nt, lambda a, c: (a + c).astype(F32), 0.0)
    mean = (ref_tree_sum(psum) / F32(hidden)).astype(F32)
    pvar = _thread_partials((x * x).astype(F32), nt, lambda a, c: (a + c).astype(F32), 0.0)
    var = ref_tree_sum(pvar)
    rstd = (F32(1.0) / np.sqrt(((var / F32(hidden)).astype(F32) + eps).astype(F32))).astype(F32)

    y = ((x - mean[:, None]).astype(F32) * rstd[:, None]).astype(F32)
    y = (y * g[None, :]).astype(F32)
    yh = y.astype(F16)
    amax_h = np.maximum(np.abs(yh).max(axis=1), F16(1e-6)).astype(F16)
    amax = amax_h.astype(F32)
    scale = (amax / F32(127.0)).astype(F32).astype(F16)
    dyn = (F32(127.0) / amax).astype(F32)
    q = rni_sat_s8((y * dyn[:, None]).astype(F32))
    s = None
    if fuse_sum:
        # fp16 per-thread accumulation: sum = h(f32(sum) + f32(yh))
        part = np.zeros((tokens, nt), F16)
        for start in range(0, hidden, nt):
            c = yh[:, start:start + nt]
            n = c.shape[1]
            part[:, :n] = (part[:, :n].astype(F32) + c.astype(F32)).astype(F32).astype(F16)
        s = ref_tree_sum(part.astype(F32)).astype(F16)
    return q, scale, s


def rms_norm(x_h, weight_h, eps: float):
    """rms_norm_kernel<half, half, false> (layernorm_kernels.cu:335-365):
    out = h( f32( h(x * rstd) ) * f32(w) )  -- `((scalar_t)(x*s_variance)) * weight`
    with c10::Half operator* (float multiply, rounded to half)."""
    x = np.asarray(x_h, dtype=F16).astype(F32)
    w = np.asarray(weight_h, dtype=F16).astype(F32)
    tokens, hidden = x.shape
    nt = min(hidden, 1024)
    ntp = ((nt + 31) // 32) * 32
    pvar = _thread_partials((x * x).astype(F32), nt, lambda a, c: (a + c).astype(F32), 0.0)
    if ntp != nt:
        pvar = np.concatenate([pvar, np.zeros((tokens, ntp - nt), F32)], axis=1)
    var = ref_tree_sum(pvar)
    rstd = (F32(1.0) / np.sqrt(((var / F32(hidden)).astype(F32) + F32(eps)).astype(F32))).astype(F32)
    t = (x * rstd[:, None]).astype(F32).astype(F16).astype(F32)
    return (t * w[None, :]).astype(F32).astype(F16)


def silu_and_mul(x_h):
    """silu_and_mul_kernel (activation_kernels.cu:10-30):
    out = h( f32( h( x / (1 + exp(-x)) ) ) * f32(y) ), input [..., 2d]."""
    x = np.asarray(x_h, dtype=F16)
    d = x.shape[-1] // 2
    a = x[..., :d].astype(F32)
    b = x[..., d:].astype(F32)
    e = np.exp((-a).astype(F32)).astype(F32)
    s = (a / (F32(1.0) + e).astype(F32)).astype(F32).astype(F16).astype(F32)
    return (s * b).astype(F32).astype(F16)
