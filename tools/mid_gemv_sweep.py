"""GPU sweep of the 64-row decode GEMV tile (M = 33..128) over library variants built with different weight-ring
depths (tune_libs/lib_ring{4,8,16}.so = -DOMNI_GEMV_RING_MB4=...).  One process per library:
    OMNI_TUNE_LIB=tune_libs/lib_ring16.so python tools/mid_gemv_sweep.py
Shapes: Llama-2-70B TP=8 shard at M = 128 (per-channel) and Llama-3-8B at M = 64 (g128); default plan and a few
(kw, sk) overrides.  HIP-graph timed, weights rotated over > MALL copies."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402

if os.environ.get("OMNI_TUNE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["OMNI_TUNE_LIB"])
from omniserve_amd.backend import qgemm_w4a8_per_chn, qgemm_w4a8_per_group  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib()
_lib.workspace(256 << 20, dev, "gemm")
cases = [(128, 7168, 8192, -1), (128, 8192, 3584, -1), (128, 1280, 8192, -1), (128, 8192, 1024, -1),
         (64, 28672, 4096, 128), (64, 4096, 14336, 128), (64, 6144, 4096, 128), (64, 4096, 4096, 128)]
only = os.environ.get("OMNI_SWEEP_OVERRIDES", "1") != "0"
for (M, N, K, group) in cases:
    copies = max(2, int(700e6 // (N * K // 2)))
    ws = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8) for _ in range(copies)]
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = sw.clone()
    sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = sa.clone()
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    if group > 0:
        s2z = torch.randint(0, 16, (K // group, N), dtype=torch.int8, device=dev)
        s2s = torch.randint(1, 16, (K // group, N), dtype=torch.int8, device=dev)
    alg = M * K + N * K // 2 + 2 * M * N

    def call(i):
        if group > 0:
            qgemm_w4a8_per_group.gemm_forward_cuda(a, ws[i], s2z, s2s, sw, sa, out)
        else:
            qgemm_w4a8_per_chn.gemm_forward_cuda(a, ws[i], sw, sa, sz, asum, out)

    plans = [(0, 0)] + ([(2, s) for s in (1, 2, 4, 8)] if only else [])
    if os.environ.get("OMNI_SWEEP_WIDE", "0") != "0":
        plans = [(0, 0)] + [(16 + w, s) for w in (4, 2, 8) for s in (1, 2, 4, 7, 8, 14, 16) if not (w == 8 and M <= 64)]
    if os.environ.get("OMNI_SWEEP_AR2", "0") != "0":
        plans = [(0, 0)] + [(4, s) for s in (1, 2, 4, 7, 8, 14)] + [(2, s) for s in (2, 4, 8)]
    for (kw, sk) in plans:
        if kw < 16 and kw == 4 and sk and (K % (sk * 4 * 128) or K // (sk * 4) < 256):
            continue
        if kw < 16 and kw == 4:
            pass
        elif kw < 16 and sk and (K % (sk * max(kw, 1) * 128) or K // (sk * max(kw, 1)) < 512):
            continue
        if kw >= 16 and (K % (sk * 256) or K // sk < 512):
            continue
        lib.omni_gemm_set_plan_override(kw, sk)
        for i in range(copies):
            call(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(copies):
                call(i)
        g.replay(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(4):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / (4 * copies) * 1e3
        print("M=%3d N=%5d K=%5d g=%3d kw=%d sk=%2d : %8.2f us  %7.1f GB/s" % (M, N, K, group, kw, sk, us, alg / us / 1e3),
              flush=True)
        del g
    lib.omni_gemm_set_plan_override(0, 0)
    del ws
