"""Runs one decode GEMV shape (default: gate_up M=16 N=28672 K=4096, per-channel) over rotating weight copies; used
under `rocprofv3 --pmc ...` to read the kernel's HBM traffic counters.
    python tools/gemv_loop.py [N K [M [deferred]]]        deferred=1: the slab-only variant (o / down at fused level 2); deferred=silu: the gate_up form of level 3"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd.backend import fused_ext, qgemm_w4a8_per_chn  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 28672
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
M = int(sys.argv[3]) if len(sys.argv) > 3 else 16
deferred = len(sys.argv) > 4 and sys.argv[4] == "1"
silu = len(sys.argv) > 4 and sys.argv[4] == "silu"     # the gate_up form of fusion level 3 (SiLU*mul epilogue + row maxima)
copies = max(4, int(700e6 // (N * K // 2)))
ws = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8) for _ in range(copies)]
a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = sw.clone()
sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = sa.clone()
out = torch.empty((M, N), dtype=torch.float16, device=dev)
slab = torch.empty((16 << 20,), dtype=torch.uint8, device=dev)
act = torch.empty((M, N // 2), dtype=torch.float16, device=dev)
amax = fused_ext.new_amax_slots(M, dev) if M <= 16 else None
for i in range(4 * copies):
    if silu:
        fused_ext.gemm_silu_per_chn(a, ws[i % copies], sw, sa, sz, asum, act, amax)
    elif deferred:
        fused_ext.gemm_partial_per_chn(a, ws[i % copies], slab)
    else:
        qgemm_w4a8_per_chn.gemm_forward_cuda(a, ws[i % copies], sw, sa, sz, asum, out)
torch.cuda.synchronize()
print("done N=%d K=%d M=%d deferred=%s copies=%d" % (N, K, M, deferred, copies))
