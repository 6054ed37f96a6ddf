"""GPU sweep of the decode GEMV plan (in-workgroup K parts kw x grid split sk) on the four Llama-3-8B
projection shapes.  Launches are captured in one HIP graph (weights rotated over > MALL copies) so the
figure is kernel time + the in-graph launch floor, not Python launch overhead.
Usage (GPU box): python tools/kw_sweep.py [M] > gpurun_out/kw_sweep.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402
from omniserve_amd.backend import qgemm_w4a8_per_chn  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = [(28672, 4096), (4096, 14336), (6144, 4096), (4096, 4096)]
lib = _lib.lib()
_lib.workspace(64 << 20, dev, "gemm")
for (N, K) in shapes:
    copies = max(2, int(700e6 // (N * K // 2)))
    ws = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8) for _ in range(copies)]
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = sw.clone()
    sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = sa.clone()
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    alg = M * K + N * K // 2 + 2 * M * N + 4 * N + 4 * M
    for kw in (1, 2, 4):
        for sk in (1, 2, 4, 7, 8, 14, 16):
            if K % (sk * kw * 64) or K // (sk * kw) < 256:
                continue
            lib.omni_gemm_set_plan_override(kw, sk)
            for i in range(copies):
                qgemm_w4a8_per_chn.gemm_forward_cuda(a, ws[i], sw, sa, sz, asum, out)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(copies):
                    qgemm_w4a8_per_chn.gemm_forward_cuda(a, ws[i], sw, sa, sz, asum, out)
            g.replay(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(4):
                g.replay()
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) / (4 * copies) * 1e3
            print("M=%d N=%d K=%d kw=%d sk=%2d : %8.2f us  %7.1f GB/s" % (M, N, K, kw, sk, us, alg / us / 1e3), flush=True)
            del g
    lib.omni_gemm_set_plan_override(0, 0)
    del ws
