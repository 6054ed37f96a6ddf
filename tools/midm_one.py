"""One mid-M GEMM shape in a loop (for rocprofv3 passes): python tools/midm_one.py M N K [chn|grp|w8] [iters] [mode_bits]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402
from omniserve_amd.backend import qgemm_w4a8_per_chn, qgemm_w4a8_per_group, qgemm_w8a8  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "chn"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 40
bits = int(sys.argv[6]) if len(sys.argv) > 6 else 1
dev = torch.device("cuda:0")
wbytes = N * K if mode == "w8" else N * K // 2
copies = max(2, min(16, int(600e6 // wbytes)))
if mode == "w8":
    ws = [torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev) for _ in range(copies)]
else:
    ws = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8) for _ in range(copies)]
a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = sw.clone()
sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = sa.clone()
s2s = torch.randint(1, 9, (K // 128, N), dtype=torch.int8, device=dev)
s2z = torch.randint(-100, 1, (K // 128, N), dtype=torch.int8, device=dev)
out = torch.empty((M, N), dtype=torch.float16, device=dev)
_lib.lib().omni_gemm_set_midm_override(bits, 0)
for i in range(iters):
    w = ws[i % copies]
    if mode == "chn":
        qgemm_w4a8_per_chn.gemm_forward_cuda(a, w, sw, sa, sz, asum, out)
    elif mode == "grp":
        qgemm_w4a8_per_group.gemm_forward_cuda(a, w, s2z, s2s, sw, sa, out)
    else:
        qgemm_w8a8.w8a8_gemm_forward_cuda(a, w, sw, sa, out)
torch.cuda.synchronize()
