"""ctypes binding of libomniserve_hip.so -- the only way the host side reaches the kernels.

There is NO CPU fallback: if the library cannot be loaded every op raises.  torch is imported
first so that the HIP runtime already mapped by PyTorch (same soname, libamdhip64.so.7) is the
one our library binds to: stream handles and device pointers are then interchangeable.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch  # noqa: F401  (must precede the dlopen below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libomniserve_hip.so")
_SHIPPED_LIB_PATH = LIB_PATH      # tools/ point LIB_PATH at tuning variants (tune_libs/*.so), possibly built from older sources

_c = ctypes
_vp, _i, _i64, _sz, _f = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_size_t, _c.c_float

# name -> (restype, argtypes); mirrors include/omniserve_hip.h one to one
PROTOTYPES = {
    "omni_abi_version": (_i, []),
    "omni_gemm_workspace_bytes": (_sz, [_i, _i, _i]),
    "omni_gemm_partial_workspace_bytes": (_sz, [_i, _i, _i]),
    "omni_gemm_set_plan_override": (None, [_i, _i]),
    "omni_gemm_set_midm_override": (None, [_i, _i]),
    "omni_prefetch_arm_gemm": (_i, [_vp, _i, _i, _i, _i, _i, _i64, _i]),
    "omni_gemm_get_plan": (None, [_i, _i, _i, _i, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i)]),
    "omni_gemm_rowfree_ok": (_i, [_i, _i, _i, _i, _i]),
    "omni_w4a8_per_chn_gemm": (_i, [_vp] * 7 + [_i, _i, _i, _i64, _vp, _sz, _vp]),
    "omni_w4a8_per_group_gemm": (_i, [_vp] * 7 + [_i, _i, _i, _i64, _vp, _sz, _vp]),
    "omni_w8a8_gemm": (_i, [_vp] * 5 + [_i, _i, _i, _i64, _vp, _sz, _vp]),
    "omni_quant": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "omni_quant_fuse_sum": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "omni_rms_norm": (_i, [_vp, _vp, _vp, _f, _i, _i, _vp]),
    "omni_rms_norm_general": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "omni_rms_norm_general_fuse_sum": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "omni_silu_and_mul": (_i, [_vp, _vp, _i, _i, _vp]),
    "omni_quant_dt": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "omni_rms_norm_general_dt": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _vp]),
    "omni_rms_norm_dt": (_i, [_vp, _vp, _vp, _f, _i, _i, _i, _vp]),
    "omni_silu_and_mul_dt": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "omni_quant_static": (_i, [_vp, _vp, _f, _i, _i, _vp]),
    "omni_dequant": (_i, [_vp, _vp, _f, _i, _i, _c.c_longlong, _c.c_longlong, _vp]),
    "omni_dequant_add_residual": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "omni_rms_norm_quant": (_i, [_vp, _vp, _vp, _f, _i, _i, _vp]),
    "omni_rms_norm_general_static": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "omni_dequant_add_residual_rms_norm_quant": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _vp]),
    "omni_gelu": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "omni_dequant_silu_and_mul_quant": (_i, [_vp, _vp, _f, _f, _f, _vp, _vp, _i, _i, _vp]),
    "omni_add_rms_norm_general_fuse_sum": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "omni_silu_mul_quant_fuse_sum": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "omni_w4a8_per_chn_gemm_partial": (_i, [_vp, _vp, _vp, _sz, _i, _i, _i, _c.POINTER(_i), _vp]),
    "omni_w4a8_per_chn_gemm_silu": (_i, [_vp] * 8 + [_i, _i, _i, _vp]),
    "omni_w4a8_per_group_gemm_silu": (_i, [_vp] * 8 + [_i, _i, _i, _vp]),
    "omni_w4a8_per_chn_gemm_partial_f16": (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _vp, _i, _i, _i, _c.POINTER(_i), _vp]),
    "omni_w4a8_per_group_gemm_partial_f16": (_i, [_vp] * 6 + [_sz, _vp, _vp, _i, _i, _i, _c.POINTER(_i), _vp]),
    "omni_kv4_decode_attention_f16_amax": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp,
                                                _sz, _vp, _sz, _vp]),
    "omni_attn_merge_f16_amax": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "omni_splitk_add_rms_norm": (_i, [_vp, _vp, _vp, _i] + [_vp] * 5 + [_f, _i, _i, _vp]),
    "omni_decode_arm_qkv_slabs": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "omni_decode_step_begin": (_i, [_vp, _vp, _vp, _i, _i, _c.c_int64, _vp, _i, _vp, _c.c_longlong, _vp]),
    "omni_w8a8_gemm_silu": (_i, [_vp] * 6 + [_i, _i, _i, _vp]),
    "omni_w8a8_gemm_partial_f16": (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _i, _i, _i, _c.POINTER(_i), _vp]),
    "omni_splitk_add_rms_norm_general_fuse_sum": (_i, [_vp, _vp, _vp, _i] + [_vp] * 7 + [_f, _i, _i, _vp]),
    "omni_w8a8_gemm_partial": (_i, [_vp, _vp, _vp, _sz, _i, _i, _i, _c.POINTER(_i), _vp]),
    "omni_w4a8_per_group_gemm_partial": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _c.POINTER(_i), _vp]),
    "omni_splitk_w8_add_rms_norm_general_fuse_sum": (_i, [_vp, _vp, _vp, _i] + [_vp] * 5 + [_f, _i, _i, _vp]),
    "omni_tp_alloc": (_i, [_sz, _c.POINTER(_vp)]),
    "omni_tp_free": (_i, [_vp]),
    "omni_tp_ipc_handle": (_i, [_vp, _vp]),
    "omni_tp_ipc_open": (_i, [_vp, _c.POINTER(_vp)]),
    "omni_tp_ipc_close": (_i, [_vp]),
    "omni_tp_allreduce_f16": (_i, [_vp, _c.POINTER(_vp), _c.POINTER(_vp), _i, _i, _c.c_longlong, _c.c_longlong, _c.c_longlong,
                                   _i, _vp]),
    "omni_tp_add_rms_norm_general_fuse_sum": (_i, [_vp, _vp, _c.POINTER(_vp), _c.POINTER(_vp), _i, _i, _c.c_longlong, _vp, _vp,
                                                   _vp, _f, _i, _i, _c.c_longlong, _i, _vp]),
    "omni_compute_padding_offsets": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "omni_kv4_prefill_write": (_i, [_vp, _vp, _vp, _vp] + [_i] * 8 + [_vp, _i, _i, _vp]),
    "omni_kv4_decode_set_split_override": (None, [_i]),
    "omni_prefill_set_variant": (None, [_i]),
    "omni_prefill_set_xcd_split": (None, [_i]),
    "omni_kv4_decode_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "omni_kv_min_max_pool": (_i, [_vp, _vp, _vp, _vp] + [_i] * 9 + [_vp]),
    "omni_kv_page_selector": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _vp] + [_i] * 10 + [_vp, _i, _vp]),
    "omni_prefill_attention": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp] + [_i] * 6 + [_vp, _vp, _vp]),
    "omni_prefill_attention_block_streaming": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp] + [_i] * 5 + [_vp, _vp, _vp]),
    "omni_kv4_decode_attention_partial": (_i, [_vp, _vp, _vp, _i64, _i64, _vp, _vp] + [_i] * 7 + [_vp, _i, _vp, _sz, _c.POINTER(_i), _vp]),
    "omni_attn_merge_quant_fuse_sum": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "omni_kv_decode_attention_fine_grained_partial": (_i, [_vp, _vp, _vp, _i64, _i64] + [_vp] * 8 + [_i] * 16 +
                                                      [_vp, _i, _vp, _sz, _c.POINTER(_i), _vp]),
    "omni_select_topk_pages": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "omni_gather_rows_f16": (_i, [_vp, _vp, _vp, _i, _i, _i64, _vp]),
    "omni_argmax_workspace_bytes": (_sz, [_i]),
    "omni_argmax_f16": (_i, [_vp, _vp, _i64, _i, _i, _vp, _sz, _vp]),
    "omni_kv4_prefill_write_fine_grained": (_i, [_vp] * 7 + [_i] * 15 + [_vp, _i, _i, _vp]),
    "omni_kv4_decode_attention_fine_grained": (_i, [_vp, _vp, _vp, _vp, _i64, _i64] + [_vp] * 6 + [_i] * 16 + [_vp, _i, _vp, _sz, _vp]),
    "omni_kv8_prefill_write_per_tensor": (_i, [_vp] * 8 + [_i] * 15 + [_vp, _i, _i, _vp]),
    "omni_kv8_decode_attention_per_tensor": (_i, [_vp, _vp, _vp, _vp, _i64, _i64] + [_vp] * 8 + [_i] * 16 + [_vp, _i, _vp, _sz, _vp]),
    "omni_norm_gemm_fused_ok": (_i, [_i, _i, _i, _i, _i]),
    "omni_w4a8_per_chn_norm_gemm_fused": (_i, [_vp, _vp, _vp, _i] + [_vp] * 8 + [_f] + [_vp] * 4 + [_c.c_longlong, _vp, _vp, _vp,
                                                _i, _i, _i, _vp, _vp]),
    "omni_w4a8_per_group_norm_gemm_fused": (_i, [_vp, _vp, _vp, _i] + [_vp] * 6 + [_f] + [_vp] * 5 + [_c.c_longlong, _vp, _vp, _vp,
                                                  _i, _i, _i, _vp, _vp]),
    "omni_kv4_decode_attention": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp] + [_i] * 7 + [_vp, _i, _vp, _sz, _vp]),
}

_lib = None
_lock = threading.Lock()


def lib() -> ctypes.CDLL:
    """Load (once) and return the library.  Raises RuntimeError if it does not exist."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    "omniserve_amd: %s is missing -- run `python -m omniserve_amd.build` "
                    "(there is no CPU fallback)" % LIB_PATH)
            h = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
            for name, (res, args) in PROTOTYPES.items():
                if LIB_PATH != _SHIPPED_LIB_PATH and not hasattr(h, name):
                    continue           # an A/B variant built before this entry point existed: calling it fails, nothing else does
                fn = getattr(h, name)  # AttributeError = header/library drift
                fn.restype = res
                fn.argtypes = args
            _lib = h
    return _lib


# The pybind11 fast path of the per-step mirror functions (csrc_ext/omni_ext.cpp, built by omniserve_amd.build next to the
# library): same checks, same errors, same C-ABI calls, ~1.5 us of host time per call instead of ~7 us of Python + ctypes --
# what an UNMODIFIED eager reference host is bound by (bench.py `drop_in`).  None when it is not built, when there is no GPU,
# when a library variant (LIB_PATH changed) or a test harness's fake C ABI is in use, or when USE_EXT is cleared (A/B, tests).
USE_EXT = True
_DEFAULT_LIB_PATH = os.path.abspath(LIB_PATH)
_ext = None
_ext_tried = False


def fast():
    global _ext, _ext_tried
    if not USE_EXT or (_lib is not None and not isinstance(_lib, ctypes.CDLL)):
        return None
    if not _ext_tried:
        _ext_tried = True
        if os.path.abspath(LIB_PATH) == _DEFAULT_LIB_PATH and torch.cuda.is_available():
            try:
                h = lib()          # the library first: the extension binds to the same loaded image
                from . import _omni_ext as e
                if int(e.abi_version()) == int(h.omni_abi_version()):
                    _ext = e
            except ImportError:
                _ext = None
    return _ext


_ERR = {-22: "invalid argument", -12: "workspace too small", -5: "kernel launch failed"}


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError("%s failed: %s (%d)" % (what, _ERR.get(rc, "error"), rc))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream() -> int:
    """Raw hipStream_t of PyTorch's current stream on the current device (the private fast accessor when this torch has
    it: the public `torch.cuda.current_stream().cuda_stream` builds a Stream object per call, ~1.5 us on a path that runs
    ten times per decoder layer)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def elem_dtype(t, what: str) -> int:
    """0 / 1 / 2 for fp16 / bf16 / fp32 tensors (the element types the reference's row kernels are dispatched over,
    kernels/csrc/dispatch_utils.h:7-14); anything else raises like AT_DISPATCH would."""
    if t.dtype == torch.float16:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    if t.dtype == torch.float32:
        return 2
    raise RuntimeError('"%s" not implemented for \'%s\'' % (what, str(t.dtype).replace("torch.", "")))


def require_cuda(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("omniserve_amd kernels need device tensors (got a %s tensor)" % t.device)


_workspaces = {}
_retired = []     # superseded scratch buffers: captured HIP graphs may still hold their raw pointers


def workspace(nbytes: int, device, tag: str = "gemm") -> torch.Tensor:
    """Persistent per-(device, tag) scratch; grows geometrically, never shrinks.
    (Pre-size it before HIP-graph capture: growing allocates.)  A superseded buffer is never released: a HIP graph
    captured earlier has its address baked in and keeps using it (geometric growth bounds the total at twice the
    final size)."""
    key = (device, tag)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        size = max(int(nbytes), 1 << 20)
        if buf is not None:
            size = max(size, 2 * buf.numel())
            _retired.append(buf)
        buf = torch.empty(size, dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf
