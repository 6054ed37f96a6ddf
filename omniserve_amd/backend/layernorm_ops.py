"""Mirror of omniserve_backend.layernorm_ops (kernels/csrc/layernorm.cpp:17-77): every function and overload."""
import torch

from .. import _lib


def _shape(input):
    hidden = input.shape[-1]
    return input.numel() // hidden, hidden


def rms_norm(out, input, weight, epsilon, use_quant=False):
    """fp16 out (layernorm_kernels.cu:335-365), or with use_quant int8 out = rni_sat((x * rstd) * w)."""
    if not use_quant and input.dtype is torch.float16 and weight.dtype is torch.float16 and out.dtype is torch.float16:
        f = _lib.fast()
        if f is not None:
            return f.rms_norm_f16(out, input, weight, float(epsilon))
    _lib.require_cuda(out, input, weight)
    tokens, hidden = _shape(input)
    fn = _lib.lib().omni_rms_norm_quant if use_quant else _lib.lib().omni_rms_norm
    if use_quant and out.dtype != torch.int8:
        raise RuntimeError("rms_norm(use_quant=True): int8 output expected")
    if use_quant and tokens == 0:
        return
    dt = _lib.elem_dtype(input, "rms_norm_kernel")      # fp16 (hot path) | bf16 | fp32, as the reference dispatches
    if weight.dtype != input.dtype:
        raise RuntimeError("rms_norm: weight and input must have the same dtype")
    if dt != 0:
        if use_quant:
            raise NotImplementedError("rms_norm(use_quant=True) is built for fp16 inputs only (no caller upstream)")
        if out.dtype != input.dtype:
            raise RuntimeError("rms_norm: out and input must have the same dtype")
        rc = _lib.lib().omni_rms_norm_dt(out.data_ptr(), input.data_ptr(), weight.data_ptr(), float(epsilon), tokens, hidden, dt,
                                         _lib.current_stream())
        return _lib.check(rc, "layernorm_ops.rms_norm")
    rc = fn(out.data_ptr(), input.data_ptr(), weight.data_ptr(), float(epsilon), tokens, hidden, _lib.current_stream())
    _lib.check(rc, "layernorm_ops.rms_norm")


def rms_norm_general(out, input, weight, scaling, epsilon, use_per_token_quant=False):
    """Per token: scaling [tokens] is written (layernorm_kernels.cu:443-454); per tensor: scaling [1] is read,
    q = rni_sat(y * scaling) (:455-466)."""
    if use_per_token_quant and input.dtype is torch.float16 and weight.dtype is torch.float16:
        f = _lib.fast()
        if f is not None:
            return f.rms_norm_general_f16(out, input, weight, scaling, float(epsilon))
    _lib.require_cuda(out, input, weight, scaling)
    tokens, hidden = _shape(input)
    fn = _lib.lib().omni_rms_norm_general if use_per_token_quant else _lib.lib().omni_rms_norm_general_static
    if tokens == 0 and not use_per_token_quant:
        return
    dt = _lib.elem_dtype(input, "generalLayerNorm")
    if weight.dtype != input.dtype:
        raise RuntimeError("rms_norm_general: weight and input must have the same dtype")
    if dt != 0:
        if not use_per_token_quant:
            raise NotImplementedError("rms_norm_general (per-tensor scale) is built for fp16 inputs only (no caller upstream)")
        rc = _lib.lib().omni_rms_norm_general_dt(out.data_ptr(), input.data_ptr(), weight.data_ptr(), None, scaling.data_ptr(),
                                                 float(epsilon), tokens, hidden, dt, _lib.current_stream())
        return _lib.check(rc, "layernorm_ops.rms_norm_general")
    rc = fn(out.data_ptr(), input.data_ptr(), weight.data_ptr(), scaling.data_ptr(), float(epsilon), tokens, hidden,
            _lib.current_stream())
    _lib.check(rc, "layernorm_ops.rms_norm_general")


def rms_norm_general_fuse_sum(out, input, weight, input_sum, scaling, epsilon, use_per_token_quant=False):
    if not use_per_token_quant:
        raise AssertionError("rms_norm_general_fuse_sum: per-tensor input_sum is not implemented "
                             "(the reference asserts here too, layernorm_kernels.cu:499-501)")
    if input.dtype is torch.float16 and weight.dtype is torch.float16:
        f = _lib.fast()
        if f is not None:
            return f.rms_norm_general_fuse_sum_f16(out, input, weight, input_sum, scaling, float(epsilon))
    _lib.require_cuda(out, input, weight, input_sum, scaling)
    tokens, hidden = _shape(input)
    dt = _lib.elem_dtype(input, "generalLayerNorm_fuse_sum")
    if weight.dtype != input.dtype:
        raise RuntimeError("rms_norm_general_fuse_sum: weight and input must have the same dtype")
    if dt != 0:
        rc = _lib.lib().omni_rms_norm_general_dt(out.data_ptr(), input.data_ptr(), weight.data_ptr(), input_sum.data_ptr(),
                                                 scaling.data_ptr(), float(epsilon), tokens, hidden, dt, _lib.current_stream())
        return _lib.check(rc, "layernorm_ops.rms_norm_general_fuse_sum")
    rc = _lib.lib().omni_rms_norm_general_fuse_sum(out.data_ptr(), input.data_ptr(), weight.data_ptr(),
                                                   input_sum.data_ptr(), scaling.data_ptr(),
                                                   float(epsilon), tokens, hidden, _lib.current_stream())
    _lib.check(rc, "layernorm_ops.rms_norm_general_fuse_sum")


def invoke_dequant_add_residual_rms_norm_quant(out, input, residual, gamma, scale, epsilon):
    """residual += int32 input * scale (in place); out int8 = rni_sat(rms_norm(residual) * gamma)
    (layernorm_kernels.cu:370-409, launches :515-561); `scale` a float or a [tokens] fp16 tensor."""
    _lib.require_cuda(out, input, residual, gamma)
    if not (input.is_contiguous() and out.is_contiguous() and residual.is_contiguous()):
        raise RuntimeError("invoke_dequant_add_residual_rms_norm_quant: tensors must be contiguous")
    if input.dtype != torch.int32 or residual.dtype != torch.float16 or out.dtype != torch.int8:
        raise RuntimeError("invoke_dequant_add_residual_rms_norm_quant: int32 input, fp16 residual, int8 out expected")
    tokens, hidden = _shape(input)
    per_token = torch.is_tensor(scale)
    if per_token:
        _lib.require_cuda(scale)
        if scale.dtype != torch.float16 or scale.numel() < tokens:
            raise RuntimeError("invoke_dequant_add_residual_rms_norm_quant: scale must be fp16 [tokens]")
    if tokens == 0:
        return
    rc = _lib.lib().omni_dequant_add_residual_rms_norm_quant(
        out.data_ptr(), input.data_ptr(), residual.data_ptr(), gamma.data_ptr(),
        scale.data_ptr() if per_token else None, 0.0 if per_token else float(scale), float(epsilon), tokens, hidden,
        _lib.current_stream())
    _lib.check(rc, "layernorm_ops.invoke_dequant_add_residual_rms_norm_quant")
