"""GPU tuning sweep for the decode-shape W4A8 GEMM plan: times every (waves, split-K) override on
the four Llama-3-8B projection shapes with HIP events, weights rotated over 8 copies (> MALL).
Usage (on the GPU box): python tools/gemv_sweep.py [M] > gpurun_out/gemv_sweep.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402
from omniserve_amd.backend import qgemm_w4a8_per_chn  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]
lib = _lib.lib()
for (N, K) in shapes:
    copies = max(2, int(600e6 // (N * K // 2)))
    ws = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8) for _ in range(copies)]
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = sw.clone()
    sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = sa.clone()
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    alg = M * K + N * K // 2 + 2 * M * N + 4 * N + 4 * M
    for waves in (4, 1):
        for sk in (1, 2, 4, 8, 16, 32):
            if K % (sk * 64) or K // sk < 256:
                continue
            lib.omni_gemm_set_plan_override(waves, sk)
            for i in range(copies):
                qgemm_w4a8_per_chn.gemm_forward_cuda(a, ws[i], sw, sa, sz, asum, out)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 5 * copies
            s.record()
            for i in range(iters):
                qgemm_w4a8_per_chn.gemm_forward_cuda(a, ws[i % copies], sw, sa, sz, asum, out)
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) / iters * 1e3
            print("M=%d N=%d K=%d waves=%d sk=%2d : %8.2f us  %7.1f GB/s" % (M, N, K, waves, sk, us, alg / us / 1e3), flush=True)
    lib.omni_gemm_set_plan_override(0, 0)
    del ws
