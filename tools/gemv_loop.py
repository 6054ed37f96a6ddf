"""Runs the gate_up decode GEMV (M=16, N=28672, K=4096, per-channel) over rotating weight copies;
used under `rocprofv3 --pmc ...` to read the kernel's HBM traffic counters."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd.backend import qgemm_w4a8_per_chn  # noqa: E402

dev = torch.device("cuda:0")
M, N, K = 16, 28672, 4096
copies = 12
ws = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8) for _ in range(copies)]
a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = sw.clone()
sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = sa.clone()
out = torch.empty((M, N), dtype=torch.float16, device=dev)
for i in range(4 * copies):
    qgemm_w4a8_per_chn.gemm_forward_cuda(a, ws[i % copies], sw, sa, sz, asum, out)
torch.cuda.synchronize()
print("done")
