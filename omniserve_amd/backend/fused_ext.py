"""Opt-in fused entry points (NOT in the reference's omniserve_backend).
Each is bit-identical to the pair of reference calls it replaces."""
from .. import _lib


def add_rms_norm_general_fuse_sum(out, residual, delta, weight, input_sum, scaling, epsilon):
    """residual += delta (in place, fp16), then rms_norm_general_fuse_sum(out, residual, ...)."""
    _lib.require_cuda(out, residual, delta, weight, input_sum, scaling)
    hidden = residual.shape[-1]
    tokens = residual.numel() // hidden
    if not residual.is_contiguous() or not delta.is_contiguous():
        raise RuntimeError("add_rms_norm_general_fuse_sum: residual and delta must be contiguous")
    rc = _lib.lib().omni_add_rms_norm_general_fuse_sum(
        out.data_ptr(), residual.data_ptr(), delta.data_ptr(), weight.data_ptr(), input_sum.data_ptr(),
        scaling.data_ptr(), float(epsilon), tokens, hidden, _lib.current_stream())
    _lib.check(rc, "fused_ext.add_rms_norm_general_fuse_sum")


def silu_mul_quant_fuse_sum(out, input, input_sum, scale):
    """silu_and_mul(tmp, input); invoke_quant_fuse_sum(out, tmp, input_sum, scale) without tmp."""
    _lib.require_cuda(out, input, input_sum, scale)
    d = input.shape[-1] // 2
    tokens = input.numel() // input.shape[-1]
    rc = _lib.lib().omni_silu_mul_quant_fuse_sum(out.data_ptr(), input.data_ptr(), input_sum.data_ptr(),
                                                 scale.data_ptr(), tokens, d, _lib.current_stream())
    _lib.check(rc, "fused_ext.silu_mul_quant_fuse_sum")


def gemm_partial_per_chn(in_feats, kernel, slab):
    """Decode-shape W4A8 per-channel GEMM without its epilogue: writes int32 partial sums
    slab[sk][M][N] and returns sk.  Pair with splitk_add_rms_norm_general_fuse_sum."""
    import ctypes
    _lib.require_cuda(in_feats, kernel, slab)
    M, K = in_feats.shape
    N = kernel.shape[0]
    sk = ctypes.c_int(0)
    rc = _lib.lib().omni_w4a8_per_chn_gemm_partial(in_feats.data_ptr(), kernel.data_ptr(), slab.data_ptr(),
                                                   slab.numel() * slab.element_size(), M, N, K,
                                                   ctypes.byref(sk), _lib.current_stream())
    _lib.check(rc, "fused_ext.gemm_partial_per_chn")
    return sk.value


def splitk_add_rms_norm_general_fuse_sum(out, residual, slab, sk, wscales, ascales_in, w_szs, a_ssums_in,
                                         weight, input_sum, scaling, epsilon):
    """residual += fp16(per-channel GEMM epilogue(sum of sk slabs)); then norm + quant (+sum) of it."""
    _lib.require_cuda(out, residual, slab, wscales, ascales_in, w_szs, a_ssums_in, weight, input_sum, scaling)
    hidden = residual.shape[-1]
    tokens = residual.numel() // hidden
    rc = _lib.lib().omni_splitk_add_rms_norm_general_fuse_sum(
        out.data_ptr(), residual.data_ptr(), slab.data_ptr(), int(sk), wscales.data_ptr(), ascales_in.data_ptr(),
        w_szs.data_ptr(), a_ssums_in.data_ptr(), weight.data_ptr(), input_sum.data_ptr(), scaling.data_ptr(),
        float(epsilon), tokens, hidden, _lib.current_stream())
    _lib.check(rc, "fused_ext.splitk_add_rms_norm_general_fuse_sum")
