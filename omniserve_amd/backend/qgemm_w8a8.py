"""Mirror of omniserve_backend.qgemm_w8a8 (kernels/csrc/qgemm/w8a8/w8a8_gemm_cuda.h:16)."""
from .. import _lib
from ._gemm_common import check_gemm_io, gemm_workspace


def w8a8_gemm_forward_cuda(in_feats, kernel, wscales, ascales, out_feats):
    f = _lib.fast()
    if f is not None:
        return f.gemm_w8a8(in_feats, kernel, wscales, ascales, out_feats)
    M, N, K, stride = check_gemm_io(in_feats, kernel, out_feats, packed=False)
    _lib.require_cuda(wscales, ascales)
    ws = gemm_workspace(M, N, K, in_feats.device)
    rc = _lib.lib().omni_w8a8_gemm(
        in_feats.data_ptr(), kernel.data_ptr(), wscales.data_ptr(), ascales.data_ptr(),
        out_feats.data_ptr(), M, N, K, stride, ws.data_ptr(), ws.numel(), _lib.current_stream())
    _lib.check(rc, "qgemm_w8a8.w8a8_gemm_forward_cuda")
