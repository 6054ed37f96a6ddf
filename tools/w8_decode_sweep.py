"""W8A8 decode GEMV (LServe, M = 1): plan sweep (waves per workgroup = K parts in the workgroup, K splits across workgroups)
for the Llama-3-8B projections, HIP-graph timed over rotating cold weight copies (each launch = GEMV + slab epilogue if split)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402
if os.environ.get("OMNI_TUNE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["OMNI_TUNE_LIB"])
from omniserve_amd.backend import qgemm_w8a8  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib()
_lib.workspace(256 << 20, dev, "gemm")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for (N, K) in [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]:
    copies = max(4, int(900e6 // (N * K)))
    ws = [torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev) for _ in range(copies)]
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev)
    sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    for (kw, sk) in [(0, 0), (4, 1), (4, 2), (4, 4), (2, 2), (2, 4), (2, 8), (1, 4), (1, 8), (1, 16), (4, 8), (2, 16)]:
        if kw and (K % (sk * kw * 64) or K // (sk * kw) < 256):
            continue
        lib.omni_gemm_set_plan_override(kw, sk)
        try:
            for i in range(copies):
                qgemm_w8a8.w8a8_gemm_forward_cuda(a, ws[i], sw, sa, out)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(copies):
                    qgemm_w8a8.w8a8_gemm_forward_cuda(a, ws[i], sw, sa, out)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                g.replay()
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) / (5 * copies) * 1e3
            print("M=%d N=%5d K=%5d kw=%d sk=%2d : %7.2f us  %7.1f GB/s  (%d workgroups)" % (
                M, N, K, kw, sk, us, (N * K + M * K + 2 * M * N) / us / 1e3, (N // 64) * max(sk, 1)), flush=True)
        except RuntimeError as ex:
            print("M=%d N=%d K=%d kw=%d sk=%d: %s" % (M, N, K, kw, sk, str(ex)[:80]))
    lib.omni_gemm_set_plan_override(0, 0)
    del ws
