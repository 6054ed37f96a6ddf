"""Mirror of omniserve_backend.fused_kernels (kernels/csrc/fused.cpp:52-76), tensor-scale overloads."""
import torch

from .. import _lib


def _shape(out, input):
    _lib.require_cuda(out, input)
    if not input.is_contiguous() or not out.is_contiguous():
        raise RuntimeError("invoke_quant: tensors must be contiguous")
    hidden = input.shape[-1]
    return input.numel() // hidden, hidden


def invoke_quant(out, input, scale):
    """Per-token int8 quantisation (fused_kernels.cu:235-250).  `scale` must be a [tokens] fp16
    tensor; the scalar (per-tensor) overload of the reference is not on the W4A8 path."""
    if not torch.is_tensor(scale):
        raise NotImplementedError("invoke_quant: per-tensor (scalar scale) overload is not implemented")
    tokens, hidden = _shape(out, input)
    if tokens == 0:
        return
    rc = _lib.lib().omni_quant(out.data_ptr(), input.data_ptr(), scale.data_ptr(), tokens, hidden,
                               _lib.current_stream())
    _lib.check(rc, "fused_kernels.invoke_quant")


def invoke_quant_fuse_sum(out, input, input_sum, scale):
    """Per-token int8 quantisation + fp16 row sum (fused_kernels.cu:255-271)."""
    if not torch.is_tensor(scale) or not torch.is_tensor(input_sum):
        raise NotImplementedError("invoke_quant_fuse_sum: scalar overload is not implemented")
    tokens, hidden = _shape(out, input)
    if tokens == 0:
        return
    rc = _lib.lib().omni_quant_fuse_sum(out.data_ptr(), input.data_ptr(), input_sum.data_ptr(),
                                        scale.data_ptr(), tokens, hidden, _lib.current_stream())
    _lib.check(rc, "fused_kernels.invoke_quant_fuse_sum")
