"""Mirror of omniserve_backend.activation_ops (kernels/csrc/activation.cpp), silu_and_mul only."""
from .. import _lib


def silu_and_mul(out, input):
    _lib.require_cuda(out, input)
    d = input.shape[-1] // 2
    tokens = input.numel() // input.shape[-1]
    rc = _lib.lib().omni_silu_and_mul(out.data_ptr(), input.data_ptr(), tokens, d, _lib.current_stream())
    _lib.check(rc, "activation_ops.silu_and_mul")
