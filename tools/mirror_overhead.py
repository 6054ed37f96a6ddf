"""Host cost of the mirror's Python marshalling per call at decode shapes (no GPU: the ctypes handle is a stub that returns 0;
add ~1.3 us of real ctypes call, the stream lookup and the HIP launch itself for the true eager cost)."""
import os
import sys
import timeit

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402
import omniserve_backend.activation_ops as act  # noqa: E402
import omniserve_backend.fused_attention_pure_dense as fa  # noqa: E402
import omniserve_backend.fused_kernels as fk  # noqa: E402
import omniserve_backend.layernorm_ops as ln  # noqa: E402
import omniserve_backend.qgemm_w4a8_per_chn as g  # noqa: E402
from omniserve_amd.backend import _attn_common  # noqa: E402


class Stub:
    def __getattr__(self, n):
        if n.endswith("workspace_bytes"):
            return lambda *a: 65536
        return lambda *a: 0


_lib._lib = Stub()
_lib.require_cuda = lambda *t: None
_lib.current_stream = lambda: 0
_rt = torch.zeros((4096, 64, 2))
_real_rope = _attn_common.rope_table
_attn_common.rope_table = lambda max_pos, dim, base, scale, device: _real_rope(max_pos, dim, base, scale, 'cpu')
B, H, I, Hq, Hk, D = 16, 4096, 14336, 32, 8, 128
i8, f16 = torch.int8, torch.float16
x8 = torch.zeros((B, H), dtype=i8); w = torch.zeros((H, H // 2), dtype=i8)
s = torch.zeros((H,), dtype=f16); sa = torch.zeros((B,), dtype=f16); out = torch.zeros((B, H), dtype=f16)
xh = torch.zeros((B, H), dtype=f16); gu = torch.zeros((B, 2 * I), dtype=f16); mid = torch.zeros((B, I), dtype=f16)
q8i = torch.zeros((B, I), dtype=i8)
qkv = torch.zeros((B, (Hq + 2 * Hk) * D), dtype=f16)
q = qkv[:, : Hq * D].view(B, Hq, D); k = qkv[:, Hq * D:(Hq + Hk) * D].view(B, Hk, D); v = qkv[:, (Hq + Hk) * D:].view(B, Hk, D)
tab = torch.zeros((B, 2, 24), dtype=torch.int64); lens = torch.zeros((B,), dtype=torch.int32)
cases = {
    "qgemm_w4a8_per_chn.gemm_forward_cuda": lambda: g.gemm_forward_cuda(x8, w, s, sa, s, sa, out),
    "layernorm_ops.rms_norm_general_fuse_sum": lambda: ln.rms_norm_general_fuse_sum(x8, xh, s, sa, sa, 1e-5, True),
    "fused_kernels.invoke_quant_fuse_sum": lambda: fk.invoke_quant_fuse_sum(x8, xh, sa, sa),
    "activation_ops.silu_and_mul": lambda: act.silu_and_mul(mid, gu),
    "fused_attention_pure_dense.single_query_attention": lambda: fa.single_query_attention(
        q, k, v, tab, lens, None, 8192, 64, Hk * D // 2, 1100, D, 500000.0, True, True, True),
}
tot = 0.0
per_layer = {"qgemm_w4a8_per_chn.gemm_forward_cuda": 4, "layernorm_ops.rms_norm_general_fuse_sum": 2,
             "fused_kernels.invoke_quant_fuse_sum": 2, "activation_ops.silu_and_mul": 1,
             "fused_attention_pure_dense.single_query_attention": 1}
for name, fn in cases.items():
    us = min(timeit.repeat(fn, number=20000, repeat=3)) / 20000 * 1e6
    tot += us * per_layer[name]
    print("%-52s %6.2f us x %d" % (name, us, per_layer[name]))
print("per layer %.1f us, x32 layers = %.2f ms of Python per decode step" % (tot, tot * 32 / 1000))
