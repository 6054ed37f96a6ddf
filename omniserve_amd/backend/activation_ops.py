"""Mirror of omniserve_backend.activation_ops (kernels/csrc/activation.cpp): silu_and_mul (the hot-path one), gelu_new,
gelu_fast and both overloads of invoke_dequant_silu_and_mul_quant."""
import torch

from .. import _lib


def silu_and_mul(out, input):
    if input.dtype is torch.float16 and out.dtype is torch.float16:
        f = _lib.fast()
        if f is not None:
            return f.silu_and_mul_f16(out, input)
    _lib.require_cuda(out, input)
    d = input.shape[-1] // 2
    tokens = input.numel() // input.shape[-1]
    dt = _lib.elem_dtype(input, "silu_and_mul_kernel")      # fp16 (hot path) | bf16 | fp32, as the reference dispatches
    if out.dtype != input.dtype:
        raise RuntimeError("silu_and_mul: out and input must have the same dtype")
    if dt != 0:
        rc = _lib.lib().omni_silu_and_mul_dt(out.data_ptr(), input.data_ptr(), tokens, d, dt, _lib.current_stream())
        return _lib.check(rc, "activation_ops.silu_and_mul")
    rc = _lib.lib().omni_silu_and_mul(out.data_ptr(), input.data_ptr(), tokens, d, _lib.current_stream())
    _lib.check(rc, "activation_ops.silu_and_mul")


def _gelu(out, input, kind, name):
    _lib.require_cuda(out, input)
    if input.dtype != torch.float16 or not input.is_contiguous() or not out.is_contiguous():
        raise RuntimeError(name + ": contiguous fp16 tensors expected")
    d = input.shape[-1]
    if input.numel() == 0:
        return
    rc = _lib.lib().omni_gelu(out.data_ptr(), input.data_ptr(), kind, input.numel() // d, d, _lib.current_stream())
    _lib.check(rc, name)


def gelu_new(out, input):
    """GPT-2 GELU in fp16 (activation_kernels.cu:186-190,200-205)."""
    _gelu(out, input, 0, "activation_ops.gelu_new")


def gelu_fast(out, input):
    """tanh-approximated GELU in fp16 (activation_kernels.cu:192-198,208-213)."""
    _gelu(out, input, 1, "activation_ops.gelu_fast")


def invoke_dequant_silu_and_mul_quant(out, input, scale_gate, scale_up, scale_out, tmp=None):
    """int32 [tokens, 2d] -> int8 [tokens, d] (activation_kernels.cu:100-131).  `scale_out` a float: static output scale;
    `scale_out` a float32 [tokens] tensor (+ `tmp` float32 [tokens, d]): per-token scale, both written."""
    _lib.require_cuda(out, input)
    if input.dtype != torch.int32 or out.dtype != torch.int8 or not input.is_contiguous() or not out.is_contiguous():
        raise RuntimeError("invoke_dequant_silu_and_mul_quant: contiguous int32 input / int8 output expected")
    d = input.shape[-1] // 2
    tokens = input.numel() // input.shape[-1]
    if tokens == 0:
        return
    if torch.is_tensor(scale_out):
        if tmp is None:
            raise TypeError("invoke_dequant_silu_and_mul_quant: the per-token overload takes (scale_out, tmp)")
        _lib.require_cuda(scale_out, tmp)
        if scale_out.dtype != torch.float32 or tmp.dtype != torch.float32 or scale_out.numel() < tokens or \
                tmp.numel() < tokens * d or not tmp.is_contiguous():
            raise RuntimeError("invoke_dequant_silu_and_mul_quant: float32 scale_out [tokens] and tmp [tokens, d] expected")
        rc = _lib.lib().omni_dequant_silu_and_mul_quant(out.data_ptr(), input.data_ptr(), float(scale_gate),
                                                        float(scale_up), 0.0, scale_out.data_ptr(), tmp.data_ptr(),
                                                        tokens, d, _lib.current_stream())
    else:
        if tmp is not None:
            raise TypeError("invoke_dequant_silu_and_mul_quant: `tmp` only goes with a tensor scale_out")
        rc = _lib.lib().omni_dequant_silu_and_mul_quant(out.data_ptr(), input.data_ptr(), float(scale_gate),
                                                        float(scale_up), float(scale_out), None, None, tokens, d,
                                                        _lib.current_stream())
    _lib.check(rc, "activation_ops.invoke_dequant_silu_and_mul_quant")
