"""Graph-timed sweep of the decode GEMV main kernel alone (deferred epilogue: slabs only) vs K splits."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib
from omniserve_amd.backend import fused_ext
from tools.sweep_graph import graph_time_us
dev = torch.device("cuda:0")
lib = _lib.lib()
M = 16
slab = torch.empty((256 << 20,), dtype=torch.uint8, device=dev)
for (N, K) in [(4096, 4096), (6144, 4096), (4096, 14336), (28672, 4096)]:
    copies = max(4, int(700e6 // (N * K // 2)))
    ws = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8) for _ in range(copies)]
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    alg = M * K + N * K // 2
    for sk in (1, 2, 4, 7, 8, 14, 16, 28, 32):
        if K % (sk * 64) or (K // sk) < 256:
            continue
        lib.omni_gemm_set_plan_override(0, sk)
        us = graph_time_us(lambda i: fused_ext.gemm_partial_per_chn(a, ws[i % copies], slab), copies)
        print("partial M=%d N=%d K=%d sk=%2d : %7.2f us  %7.1f GB/s" % (M, N, K, sk, us, alg / us / 1e3), flush=True)
    lib.omni_gemm_set_plan_override(0, 0)
    del ws
