"""Mirror of omniserve_backend.fused_kernels (kernels/csrc/fused.cpp:16-76): every function and overload of the
module.  The tensor-scale quantisers are the ones on the W4A8 / W8A8 hot path; the `at::Half scale` overloads and the
int32 dequantisers are not called by the reference's model code (csrc/offpath.hip)."""
import torch

from .. import _lib


def _shape(out, input):
    _lib.require_cuda(out, input)
    if not input.is_contiguous() or not out.is_contiguous():
        raise RuntimeError("invoke_quant: tensors must be contiguous")
    hidden = input.shape[-1]
    return input.numel() // hidden, hidden


def _quant_static(out, input, scale, name):
    tokens, hidden = _shape(out, input)
    if tokens == 0:
        return
    rc = _lib.lib().omni_quant_static(out.data_ptr(), input.data_ptr(), float(scale), tokens, hidden,
                                      _lib.current_stream())
    _lib.check(rc, name)


def invoke_quant(out, input, scale):
    """int8 quantisation: `scale` a [tokens] fp16 tensor -> per token, scale written (fused_kernels.cu:218-233);
    a Python float (at::Half) -> static, q = rni_sat(x / scale) (fused_kernels.cu:202-216)."""
    if not torch.is_tensor(scale):
        return _quant_static(out, input, scale, "fused_kernels.invoke_quant")
    if input.dtype is torch.float16:
        f = _lib.fast()
        if f is not None:
            return f.quant_f16(out, input, scale)
    tokens, hidden = _shape(out, input)
    if tokens == 0:
        return
    dt = _lib.elem_dtype(input, "quant_kernel")       # fp16 (hot path) | bf16 | fp32, as the reference dispatches
    if dt != 0:
        rc = _lib.lib().omni_quant_dt(out.data_ptr(), input.data_ptr(), None, scale.data_ptr(), tokens, hidden, dt,
                                      _lib.current_stream())
        return _lib.check(rc, "fused_kernels.invoke_quant")
    rc = _lib.lib().omni_quant(out.data_ptr(), input.data_ptr(), scale.data_ptr(), tokens, hidden,
                               _lib.current_stream())
    _lib.check(rc, "fused_kernels.invoke_quant")


def invoke_quant_fuse_sum(out, input, input_sum, scale):
    """Per-token int8 quantisation + fp16 row sum (fused_kernels.cu:255-271); with two Python floats the static
    quantiser (fused_kernels.cu:238-253: its scalar `input_sum` is never used)."""
    if not torch.is_tensor(scale) and not torch.is_tensor(input_sum):
        return _quant_static(out, input, scale, "fused_kernels.invoke_quant_fuse_sum")
    if not torch.is_tensor(scale) or not torch.is_tensor(input_sum):
        raise TypeError("invoke_quant_fuse_sum: input_sum and scale must both be tensors or both be floats")
    if input.dtype is torch.float16:
        f = _lib.fast()
        if f is not None:
            return f.quant_fuse_sum_f16(out, input, input_sum, scale)
    tokens, hidden = _shape(out, input)
    if tokens == 0:
        return
    dt = _lib.elem_dtype(input, "quant_kernel_fuse_sum")
    if dt != 0:
        rc = _lib.lib().omni_quant_dt(out.data_ptr(), input.data_ptr(), input_sum.data_ptr(), scale.data_ptr(), tokens,
                                      hidden, dt, _lib.current_stream())
        return _lib.check(rc, "fused_kernels.invoke_quant_fuse_sum")
    rc = _lib.lib().omni_quant_fuse_sum(out.data_ptr(), input.data_ptr(), input_sum.data_ptr(),
                                        scale.data_ptr(), tokens, hidden, _lib.current_stream())
    _lib.check(rc, "fused_kernels.invoke_quant_fuse_sum")


def invoke_dequant(out, input, scale):
    """out fp16 = int32 input * scale (fused_kernels.cu:184-200); rows may be strided views."""
    _lib.require_cuda(out, input)
    if input.dtype != torch.int32 or out.dtype != torch.float16:
        raise RuntimeError("invoke_dequant: int32 input and fp16 output expected")
    hidden = input.shape[-1]
    tokens = input.numel() // hidden
    if input.dim() < 2 or input.stride(-1) != 1 or out.stride(-1) != 1:
        raise RuntimeError("invoke_dequant: [..., hidden] tensors with a dense last dimension expected")
    if tokens == 0:
        return
    rc = _lib.lib().omni_dequant(out.data_ptr(), input.data_ptr(), float(scale), tokens, hidden, input.stride(-2),
                                 out.stride(-2), _lib.current_stream())
    _lib.check(rc, "fused_kernels.invoke_dequant")


def invoke_dequant_add_residual(out, input, residual, scale):
    """out = int32 input * scale + residual (fused_kernels.cu:145-182); `scale` a float or a [tokens] fp16 tensor."""
    _lib.require_cuda(out, input, residual)
    if not (input.is_contiguous() and out.is_contiguous() and residual.is_contiguous()):
        raise RuntimeError("invoke_dequant_add_residual: tensors must be contiguous")
    if input.dtype != torch.int32 or out.dtype != torch.float16 or residual.dtype != torch.float16:
        raise RuntimeError("invoke_dequant_add_residual: int32 input, fp16 residual / output expected")
    hidden = input.shape[-1]
    tokens = input.numel() // hidden
    per_token = torch.is_tensor(scale)
    if per_token:
        _lib.require_cuda(scale)
        if scale.dtype != torch.float16 or scale.numel() < tokens:
            raise RuntimeError("invoke_dequant_add_residual: scale must be fp16 [tokens]")
    if tokens == 0:
        return
    rc = _lib.lib().omni_dequant_add_residual(out.data_ptr(), input.data_ptr(), residual.data_ptr(),
                                              scale.data_ptr() if per_token else None,
                                              0.0 if per_token else float(scale), tokens, hidden,
                                              _lib.current_stream())
    _lib.check(rc, "fused_kernels.invoke_dequant_add_residual")
