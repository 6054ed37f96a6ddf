"""Mirror of omniserve_backend.layernorm_ops (kernels/csrc/layernorm.cpp:52-77)."""
from .. import _lib


def _shape(input):
    hidden = input.shape[-1]
    return input.numel() // hidden, hidden


def rms_norm(out, input, weight, epsilon, use_quant=False):
    if use_quant:
        raise NotImplementedError("rms_norm(use_quant=True) is not on the W4A8 path")
    _lib.require_cuda(out, input, weight)
    tokens, hidden = _shape(input)
    rc = _lib.lib().omni_rms_norm(out.data_ptr(), input.data_ptr(), weight.data_ptr(), float(epsilon),
                                  tokens, hidden, _lib.current_stream())
    _lib.check(rc, "layernorm_ops.rms_norm")


def rms_norm_general(out, input, weight, scaling, epsilon, use_per_token_quant=False):
    if not use_per_token_quant:
        raise NotImplementedError("rms_norm_general: per-tensor scaling is not on the W4A8 path")
    _lib.require_cuda(out, input, weight, scaling)
    tokens, hidden = _shape(input)
    rc = _lib.lib().omni_rms_norm_general(out.data_ptr(), input.data_ptr(), weight.data_ptr(),
                                          scaling.data_ptr(), float(epsilon), tokens, hidden,
                                          _lib.current_stream())
    _lib.check(rc, "layernorm_ops.rms_norm_general")


def rms_norm_general_fuse_sum(out, input, weight, input_sum, scaling, epsilon, use_per_token_quant=False):
    if not use_per_token_quant:
        raise AssertionError("rms_norm_general_fuse_sum: per-tensor input_sum is not implemented "
                             "(the reference asserts here too, layernorm_kernels.cu:499-501)")
    _lib.require_cuda(out, input, weight, input_sum, scaling)
    tokens, hidden = _shape(input)
    rc = _lib.lib().omni_rms_norm_general_fuse_sum(out.data_ptr(), input.data_ptr(), weight.data_ptr(),
                                                   input_sum.data_ptr(), scaling.data_ptr(),
                                                   float(epsilon), tokens, hidden, _lib.current_stream())
    _lib.check(rc, "layernorm_ops.rms_norm_general_fuse_sum")
