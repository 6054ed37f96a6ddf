"""Mirror of omniserve_backend.qgemm_w4a8_per_group (kernels/csrc/qgemm/w4a8_per_group/gemm_cuda.h:16)."""
from .. import _lib
from ._gemm_common import check_gemm_io, gemm_workspace


def gemm_forward_cuda(in_feats, kernel, zeros, scales_i8, wscales, ascales, out_feats):
    """g128 W4A8 GEMM; argument order as the reference (caller: w4a8_linear.py:126-135)."""
    f = _lib.fast()
    if f is not None:
        return f.gemm_w4a8_per_group(in_feats, kernel, zeros, scales_i8, wscales, ascales, out_feats)
    M, N, K, stride = check_gemm_io(in_feats, kernel, out_feats, packed=True)
    _lib.require_cuda(zeros, scales_i8, wscales, ascales)
    if tuple(zeros.shape) != (K // 128, N) or tuple(scales_i8.shape) != (K // 128, N):
        raise RuntimeError("per-group gemm: zeros/scales_i8 must be [K/128, N]")
    ws = gemm_workspace(M, N, K, in_feats.device)
    rc = _lib.lib().omni_w4a8_per_group_gemm(
        in_feats.data_ptr(), kernel.data_ptr(), zeros.data_ptr(), scales_i8.data_ptr(),
        wscales.data_ptr(), ascales.data_ptr(), out_feats.data_ptr(), M, N, K, stride,
        ws.data_ptr(), ws.numel(), _lib.current_stream())
    _lib.check(rc, "qgemm_w4a8_per_group.gemm_forward_cuda")
