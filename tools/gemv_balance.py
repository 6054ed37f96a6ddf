"""Does the M = 16 decode GEMV care how evenly its workgroups divide over the 256 CUs?  One 256-thread workgroup per
64-channel group (K split over its four waves): N / 64 workgroups.  HIP-graph timed over rotating weight copies (cold)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402
if os.environ.get("OMNI_TUNE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["OMNI_TUNE_LIB"])
from omniserve_amd.backend import qgemm_w4a8_per_chn  # noqa: E402

dev = torch.device("cuda:0")
K, M = 4096, 16
for N in (16384, 24576, 28672, 32768, 40960, 49152, 65536):
    copies = max(4, int(900e6 // (N * K // 2)))
    ws = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8) for _ in range(copies)]
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = sw.clone()
    sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = sa.clone()
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    for i in range(copies):
        qgemm_w4a8_per_chn.gemm_forward_cuda(a, ws[i], sw, sa, sz, asum, out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(copies):
            qgemm_w4a8_per_chn.gemm_forward_cuda(a, ws[i], sw, sa, sz, asum, out)
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
    alg = N * K // 2 + M * K + 2 * M * N
    print("N=%6d (%4d workgroups = %.2f per CU): %6.2f us  %6.1f GB/s" % (N, N // 64, N / 64 / 256, us, alg / us / 1e3), flush=True)
    del ws
