"""Mirror of omniserve_backend.qgemm_w4a8_per_chn (kernels/csrc/qgemm/w4a8_per_chn/gemm_cuda.h:16)."""
from .. import _lib
from ._gemm_common import check_gemm_io, gemm_workspace


def gemm_forward_cuda(in_feats, kernel, wscales, ascales, w_szs, a_ssums, out_feats):
    """out_feats[m,n] = fp16(acc*wscales[n]*ascales[m] - w_szs[n]*a_ssums[m]); writes in place,
    returns None (caller: w4a8_linear.py:111-120)."""
    f = _lib.fast()
    if f is not None:
        return f.gemm_w4a8_per_chn(in_feats, kernel, wscales, ascales, w_szs, a_ssums, out_feats)
    M, N, K, stride = check_gemm_io(in_feats, kernel, out_feats, packed=True)
    _lib.require_cuda(wscales, ascales, w_szs, a_ssums)
    ws = gemm_workspace(M, N, K, in_feats.device)
    rc = _lib.lib().omni_w4a8_per_chn_gemm(
        in_feats.data_ptr(), kernel.data_ptr(), wscales.data_ptr(), ascales.data_ptr(),
        w_szs.data_ptr(), a_ssums.data_ptr(), out_feats.data_ptr(), M, N, K, stride,
        ws.data_ptr(), ws.numel(), _lib.current_stream())
    _lib.check(rc, "qgemm_w4a8_per_chn.gemm_forward_cuda")
