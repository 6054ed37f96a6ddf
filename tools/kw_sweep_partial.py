"""Graph-timed sweep of (decode GEMV partial [kw x sk] + slab-consuming add+norm+quant kernel) pairs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib
from omniserve_amd.backend import fused_ext
from tools.sweep_graph import graph_time_us
dev = torch.device("cuda:0")
lib = _lib.lib()
M = 16
slab = torch.empty((64 << 20,), dtype=torch.uint8, device=dev)
for (N, K) in [(4096, 4096), (4096, 14336)]:
    copies = max(4, int(700e6 // (N * K // 2)))
    ws = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8) for _ in range(copies)]
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = sw.clone()
    sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = sa.clone()
    res = torch.randn((M, N), dtype=torch.float16, device=dev)
    gamma = torch.ones((N,), dtype=torch.float16, device=dev)
    outq = torch.empty((M, N), dtype=torch.int8, device=dev)
    osum = torch.empty((M,), dtype=torch.float16, device=dev); oscale = torch.empty((M,), dtype=torch.float16, device=dev)
    for kw in (1, 2, 4):
        for sk in (1, 2, 4, 7, 8, 14):
            if K % (sk * kw * 64) or (K // (sk * kw)) < 256:
                continue
            lib.omni_gemm_set_plan_override(kw, sk)

            def pair(i):
                s = fused_ext.gemm_partial_per_chn(a, ws[i % copies], slab)
                fused_ext.splitk_add_rms_norm_general_fuse_sum(outq, res, slab, s, sw, sa, sz, asum, gamma, osum, oscale, 1e-5)

            def only(i):
                fused_ext.gemm_partial_per_chn(a, ws[i % copies], slab)

            us = graph_time_us(pair, copies)
            us1 = graph_time_us(only, copies)
            print("N=%d K=%d kw=%d sk=%2d : pair %7.2f us   gemv alone %7.2f us" % (N, K, kw, sk, us, us1), flush=True)
    lib.omni_gemm_set_plan_override(0, 0)
    del ws
