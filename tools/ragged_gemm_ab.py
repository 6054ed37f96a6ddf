"""Ragged-shape prefill GEMMs (the generic 128 x 256 tile) and the 64-row GEMV tiles across library builds (OMNI_TUNE_LIB): the
kernels round 6 changed to remove spills.  python tools/ragged_gemm_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402

if os.environ.get("OMNI_TUNE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["OMNI_TUNE_LIB"])
    _lib.USE_EXT = False
from bench import event_time_ms  # noqa: E402
from omniserve_amd.backend import qgemm_w4a8_per_chn, qgemm_w4a8_per_group, qgemm_w8a8  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
name = os.path.basename(os.environ.get("OMNI_TUNE_LIB", "shipped"))
for mode, (M, N, K) in [("w8", (4100, 4096, 4096)), ("w8", (16390, 6144, 4096)), ("w8", (8200, 4096, 14336)),
                        ("chn", (64, 6144, 4096)), ("chn", (64, 4096, 4096)), ("grp", (64, 6144, 4096)), ("grp", (64, 4096, 4096))]:
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
    sw = (torch.rand((N,), device=dev, generator=g) * 0.02 + 0.001).half()
    sa = (torch.rand((M,), device=dev, generator=g) * 0.02 + 0.001).half()
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    if mode == "w8":
        w = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
        fn = lambda i: qgemm_w8a8.w8a8_gemm_forward_cuda(a, w, sw, sa, out)  # noqa: E731
    else:
        w = torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev, generator=g).view(torch.int8)
        sz = (torch.rand((N,), device=dev, generator=g) * 0.1).half()
        asum = (a.float().sum(1) * sa.float()).half()
        if mode == "chn":
            fn = lambda i: qgemm_w4a8_per_chn.gemm_forward_cuda(a, w, sw, sa, sz, asum, out)  # noqa: E731
        else:
            s2s = torch.randint(1, 16, (K // 128, N), dtype=torch.uint8, device=dev, generator=g).view(torch.int8)
            s2z = torch.randint(0, 120, (K // 128, N), dtype=torch.uint8, device=dev, generator=g).view(torch.int8)
            fn = lambda i: qgemm_w4a8_per_group.gemm_forward_cuda(a, w, s2z, s2s, sw, sa, out)  # noqa: E731
    ms = event_time_ms(fn, iters=20)
    chk = int((out.view(torch.int16).to(torch.int64) & 0xFFFF).sum().item())
    print("%-14s %-4s M=%-6d N=%-6d K=%-6d %.4f ms  %8.1f TOPS  chk %d" % (name, mode, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, chk), flush=True)
