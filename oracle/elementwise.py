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


# Element types the reference dispatches its row kernels over (kernels/csrc/dispatch_utils.h:7-14).  bf16 arrays travel as
# uint16 bit patterns (numpy has no bfloat16); `_load` widens a T array to float32 (exact), `_rt` rounds a float32 array to T
# (round to nearest even) and returns it widened again, `_store` gives the T array a kernel would have written.
def _round_bf16_bits(f: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(f, dtype=F32).view(np.uint32).astype(np.uint64)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    return np.where(nan, ((u >> 16) | 0x40).astype(np.uint16), r)


def _load(x, dtype: str) -> np.ndarray:
    if dtype == "f16":
        return np.asarray(x, dtype=F16).astype(F32)
    if dtype == "bf16":
        return (np.asarray(x, dtype=np.uint16).astype(np.uint32) << 16).view(F32)
    if dtype == "f32":
        return np.asarray(x, dtype=F32)
    raise ValueError(dtype)


def _rt(f: np.ndarray, dtype: str) -> np.ndarray:
    f = np.asarray(f, dtype=F32)
    if dtype == "f16":
        return f.astype(F16).astype(F32)
    if dtype == "bf16":
        return (_round_bf16_bits(f).astype(np.uint32) << 16).view(F32)
    return f


def _store(f: np.ndarray, dtype: str) -> np.ndarray:
    f = np.asarray(f, dtype=F32)
    if dtype == "f16":
        return f.astype(F16)
    if dtype == "bf16":
        return _round_bf16_bits(f)
    return f


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


def quant_per_token(x_h: np.ndarray, fuse_sum: bool, dtype: str = "f16"):
    """invoke_quant / invoke_quant_fuse_sum, tensor-scale overloads.

    x fp16 [tokens, hidden] -> (q int8, scale fp16[tokens], sum fp16[tokens] | None).
    block = min(hidden,1024); amax starts at 0 (no floor: an all-zero row
    divides by zero exactly as the reference does); scale = h(amax/127);
    q = rni_sat(x * (127/amax)); sum = h(tree_sum(per-thread f32 sums)).
    `dtype` = element type of x ("f16" | "bf16" as uint16 bits | "f32"): it only enters through the load; scale / sum are fp16.
    """
    x = _load(x_h, dtype)
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


def rms_norm_general(x_h, gamma_h, eps: float, fuse_sum: bool, dtype: str = "f16"):
    """generalLayerNorm[_fuse_sum]<half, at::Half>, per-token path, use_shmem=false.

    y = (x - mean) * rsqrt(mean(x^2) + eps) * gamma in f32  [mean IS subtracted
    in the output but NOT in the variance: layernorm_kernels.cu:26-34,110-113,127];
    yh = h(y); amax = max(1e-6 (as fp16), |yh|) in fp16; per-thread sum
    accumulates yh in fp16 (`T_scalar sum`, :280,291) then the f32 block tree;
    q = rni_sat( y_f32 * (127/amax) ) recomputed from f32 (:308-318);
    scale = h(amax/127).   block = roundup32(min(hidden,1024)).
    For T = bf16 / float (`dtype`) the roundings marked h() above that belong to T -- y, the running maximum, the per-thread
    running sum (`T_scalar amax`, `T_scalar sum`) -- are roundings to T; scale and sum outputs stay fp16 (at::Half).
    """
    x = _load(x_h, dtype)
    g = _load(gamma_h, dtype)
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
    yh = _rt(y, dtype)                       # float32 holding T values
    amax = np.maximum(np.abs(yh).max(axis=1), _rt(np.asarray([1e-6], F32), dtype)[0]).astype(F32)
    scale = (amax / F32(127.0)).astype(F32).astype(F16)
    dyn = (F32(127.0) / amax).astype(F32)
    q = rni_sat_s8((y * dyn[:, None]).astype(F32))
    s = None
    if fuse_sum:
        # per-thread accumulation in T: sum = T(f32(sum) + f32(yh))
        part = np.zeros((tokens, nt), F32)
        for start in range(0, hidden, nt):
            c = yh[:, start:start + nt]
            n = c.shape[1]
            part[:, :n] = _rt((part[:, :n] + c).astype(F32), dtype)
        s = ref_tree_sum(part).astype(F16)
    return q, scale, s


def rms_norm(x_h, weight_h, eps: float, dtype: str = "f16"):
    """rms_norm_kernel<half, half, false> (layernorm_kernels.cu:335-365):
    out = h( f32( h(x * rstd) ) * f32(w) )  -- `((scalar_t)(x*s_variance)) * weight`
    with c10::Half operator* (float multiply, rounded to half).  `dtype`: the same with T = bf16 (uint16 bits) / float."""
    x = _load(x_h, dtype)
    w = _load(weight_h, dtype)
    tokens, hidden = x.shape
    nt = min(hidden, 1024)
    ntp = ((nt + 31) // 32) * 32
    pvar = _thread_partials((x * x).astype(F32), nt, lambda a, c: (a + c).astype(F32), 0.0)
    if ntp != nt:
        pvar = np.concatenate([pvar, np.zeros((tokens, ntp - nt), F32)], axis=1)
    var = ref_tree_sum(pvar)
    rstd = (F32(1.0) / np.sqrt(((var / F32(hidden)).astype(F32) + F32(eps)).astype(F32))).astype(F32)
    t = _rt((x * rstd[:, None]).astype(F32), dtype)
    return _store((t * w[None, :]).astype(F32), dtype)


def silu_and_mul(x_h, dtype: str = "f16"):
    """silu_and_mul_kernel (activation_kernels.cu:10-30):
    out = h( f32( h( x / (1 + exp(-x)) ) ) * f32(y) ), input [..., 2d]."""
    x = _load(x_h, dtype)
    d = x.shape[-1] // 2
    a = x[..., :d]
    b = x[..., d:]
    e = np.exp((-a).astype(F32)).astype(F32)
    s = _rt((a / (F32(1.0) + e).astype(F32)).astype(F32), dtype)
    return _store((s * b).astype(F32), dtype)


# ------------------------------------------------------------------------------------------------------------
# Overloads the Llama W4A8 / W8A8 paths never call (SURVEY.md 8b lists them as part of the module surface):
# static (per-tensor) scales, int32 -> fp16 dequantisation, the T5-style fused residual norm, GELU.
# nvcc contracts `a * b + c` into one FMA (default -fmad=true); `_fma32` models that with a float64 product and
# sum (exact for these operand widths, then one rounding to float32).  Divisions / rsqrt / exp / tanh are IEEE
# here, approximate under the reference's --use_fast_math: compare with a tolerance where they are involved.
# ------------------------------------------------------------------------------------------------------------
def _fma32(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(F32)


def _h(scale) -> np.float32:
    """A Python float bound to an `at::Half` parameter: rounded to fp16 once (pybind11 caster)."""
    return F32(F16(scale))


def quant_static(x_h, scale: float):
    """invoke_quant / invoke_quant_fuse_sum, `at::Half scale` overloads (fused_kernels.cu:88-93,136-141,202-216,
    238-253): q = rni_sat(f32(x) / f32(scale)); the scalar `input_sum` of the fuse_sum overload is unused."""
    x = np.asarray(x_h, dtype=F16).astype(F32)
    return rni_sat_s8((x / _h(scale)).astype(F32))


def dequant(x_i32, scale: float):
    """invoke_dequant (fused_kernels.cu:45-55,184-200): out = h(f32(acc) * f32(scale))."""
    return (np.asarray(x_i32, np.int32).astype(F32) * _h(scale)).astype(F32).astype(F16)


def dequant_add_residual(x_i32, residual_h, scale):
    """invoke_dequant_add_residual (fused_kernels.cu:24-43,145-182): out = h(fma(f32(acc), f32(scale), f32(res)));
    `scale` is a Python float (at::Half) or a [tokens] fp16 array (per-token)."""
    acc = np.asarray(x_i32, np.int32).astype(F32)
    res = np.asarray(residual_h, F16).astype(F32)
    s = np.asarray(scale, F16).astype(F32).reshape(-1, 1) if np.ndim(scale) else _h(scale)
    return _fma32(acc, s, res).astype(F16)


def rms_norm_quant(x_h, weight_h, eps: float):
    """rms_norm_kernel<half, int8_t, true> (layernorm_kernels.cu:335-365): q = rni_sat((x * rstd) * f32(w)),
    both products in f32; block = min(hidden, 1024)."""
    x = np.asarray(x_h, dtype=F16).astype(F32)
    w = np.asarray(weight_h, dtype=F16).astype(F32)
    tokens, hidden = x.shape
    nt = min(hidden, 1024)
    assert nt % 32 == 0
    pvar = _thread_partials((x * x).astype(F32), nt, lambda a, c: (a + c).astype(F32), 0.0)   # x*x is exact in f32
    var = ref_tree_sum(pvar)
    rstd = (F32(1.0) / np.sqrt(((var / F32(hidden)).astype(F32) + F32(eps)).astype(F32))).astype(F32)
    return rni_sat_s8(((x * rstd[:, None]).astype(F32) * w[None, :]).astype(F32))


def rms_norm_general_static(x_h, gamma_h, scale_h, eps: float):
    """generalLayerNorm<half, at::Half>, per-tensor path (layernorm_kernels.cu:58-196, launch :455-466):
    y = h(((x - mean) * rstd) * gamma) [mean subtracted in the output only], q = rni_sat(f32(y) * f32(scale[0]))."""
    x = np.asarray(x_h, dtype=F16).astype(F32)
    g = np.asarray(gamma_h, dtype=F16).astype(F32)
    tokens, hidden = x.shape
    nt = ((min(hidden, 1024) + 31) // 32) * 32
    psum = _thread_partials(x, nt, lambda a, c: (a + c).astype(F32), 0.0)
    mean = (ref_tree_sum(psum) / F32(hidden)).astype(F32)
    pvar = _thread_partials((x * x).astype(F32), nt, lambda a, c: (a + c).astype(F32), 0.0)
    var = ref_tree_sum(pvar)
    rstd = (F32(1.0) / np.sqrt(((var / F32(hidden)).astype(F32) + F32(eps)).astype(F32))).astype(F32)
    y = ((x - mean[:, None]).astype(F32) * rstd[:, None]).astype(F32)
    yh = (y * g[None, :]).astype(F32).astype(F16)
    s = np.asarray(scale_h, F16).reshape(-1)[0].astype(F32)
    return rni_sat_s8((yh.astype(F32) * s).astype(F32))


def dequant_add_residual_rms_norm_quant(x_i32, residual_h, gamma_h, scale, eps: float):
    """invoke_dequant_add_residual_rms_norm_quant (layernorm_kernels.cu:370-409, launches :515-561):
    diff = fma(f32(acc), f32(scale), f32(res)) kept in f32 for the variance (per-thread fma(diff, diff, sum), block
    tree), residual <- h(diff) in place, q = rni_sat((f32(h(diff)) * rstd) * f32(gamma)); no mean subtraction.
    Returns (q, new residual).  block = min(hidden, 1024)."""
    acc = np.asarray(x_i32, np.int32).astype(F32)
    res = np.asarray(residual_h, F16).astype(F32)
    g = np.asarray(gamma_h, dtype=F16).astype(F32)
    tokens, hidden = acc.shape
    nt = min(hidden, 1024)
    assert nt % 32 == 0
    s = np.asarray(scale, F16).astype(F32).reshape(-1, 1) if np.ndim(scale) else _h(scale)
    diff = _fma32(acc, s, res)
    new_res = diff.astype(F16)
    pvar = _thread_partials(diff, nt, lambda a, c: _fma32(c, c, a), 0.0)
    var = ref_tree_sum(pvar)
    rstd = (F32(1.0) / np.sqrt(((var / F32(hidden)).astype(F32) + F32(eps)).astype(F32))).astype(F32)
    q = rni_sat_s8(((new_res.astype(F32) * rstd[:, None]).astype(F32) * g[None, :]).astype(F32))
    return q, new_res


def _hmul(a, b):
    with np.errstate(over="ignore"):            # fp16 overflow to inf is the reference's behaviour too
        return (a.astype(F32) * b.astype(F32)).astype(F32).astype(F16)


def _hadd(a, b):
    with np.errstate(over="ignore"):
        return (a.astype(F32) + b.astype(F32)).astype(F32).astype(F16)


def gelu_new(x_h):
    """gelu_new_kernel<c10::Half> (activation_kernels.cu:186-190): every c10::Half operator computes in f32 and
    rounds to fp16."""
    x = np.asarray(x_h, F16)
    x3 = _hmul(_hmul(x, x), x).astype(F32)
    inner = _hadd(x, (F32(0.044715) * x3).astype(F32).astype(F16))
    u = (F32(0.79788456) * inner.astype(F32)).astype(F32).astype(F16)
    t = np.tanh(u.astype(F32)).astype(F32).astype(F16)
    return _hmul(_hmul(np.full_like(x, 0.5), x), _hadd(np.full_like(x, 1.0), t))


def gelu_fast(x_h):
    """gelu_fast_kernel<c10::Half> (activation_kernels.cu:192-198)."""
    x = np.asarray(x_h, F16)
    f = x.astype(F32)
    a = (f * F32(0.79788456)).astype(F32).astype(F16)
    b = (F32(0.044715) * f).astype(F32).astype(F16)
    d = _hadd(np.full_like(x, 1.0), _hmul(b, x))
    t = np.tanh(_hmul(a, d).astype(F32)).astype(F32).astype(F16)
    return _hmul(_hmul(np.full_like(x, 0.5), x), _hadd(np.full_like(x, 1.0), t))


def dequant_silu_and_mul_quant(x_i32, scale_gate: float, scale_up: float, scale_out=None):
    """invoke_dequant_silu_and_mul_quant (activation_kernels.cu:31-86,100-131), float scales.
    t = silu(f32(gate) * scale_gate) * (f32(up) * scale_up) in f32.
    scale_out given (static): q = rni_sat(t / scale_out) -> q.
    scale_out None (per token): tmp = t, scale = amax/127 (f32), q = rni_sat((127/amax) * tmp) -> (q, scale, tmp)."""
    acc = np.asarray(x_i32, np.int32)
    d = acc.shape[-1] // 2
    x = (acc[..., :d].astype(F32) * F32(scale_gate)).astype(F32)
    y = (acc[..., d:].astype(F32) * F32(scale_up)).astype(F32)
    e = np.exp((-x).astype(F32)).astype(F32)
    t = ((x / (F32(1.0) + e).astype(F32)).astype(F32) * y).astype(F32)
    if scale_out is not None:
        return rni_sat_s8((t / F32(scale_out)).astype(F32))
    amax = np.abs(t).max(axis=-1).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = rni_sat_s8(((F32(127.0) / amax).astype(F32)[..., None] * t).astype(F32))
    return q, (amax / F32(127.0)).astype(F32), t
