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
