"""qkv projection at decode (M = 16, N = 6144, K = 4096), weights L2-resident as inside the step (the row kernel in front
prefetches them): the one-kernel form on 96 workgroups vs slab-only forms split over more workgroups (which would need the
attention kernel to apply the epilogue).  Graph-timed, the same weight buffer every launch (12.6 MB: stays in the L2s)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib
from omniserve_amd.backend import fused_ext, qgemm_w4a8_per_chn
from tools.sweep_graph import graph_time_us
dev = torch.device("cuda:0")
lib = _lib.lib()
M = 16
slab = torch.empty((64 << 20,), dtype=torch.uint8, device=dev)
for (N, K) in [(6144, 4096), (4096, 4096)]:
    w = torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8)
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = sw.clone()
    sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = sa.clone()
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    us = graph_time_us(lambda i: qgemm_w4a8_per_chn.gemm_forward_cuda(a, w, sw, sa, sz, asum, out), 32)
    print("N=%d K=%d one kernel (default plan)          : %6.2f us" % (N, K, us), flush=True)
    for kw in (4, 2, 1):
        for sk in (1, 2, 4, 8):
            if K % (sk * kw * 64) or (K // (sk * kw)) < 256:
                continue
            lib.omni_gemm_set_plan_override(kw, sk)
            us1 = graph_time_us(lambda i: fused_ext.gemm_partial_per_chn(a, w, slab), 32)
            print("N=%d K=%d slab only kw=%d sk=%d (%4d workgroups): %6.2f us" % (N, K, kw, sk, N // 64 * sk, us1), flush=True)
    lib.omni_gemm_set_plan_override(0, 0)
