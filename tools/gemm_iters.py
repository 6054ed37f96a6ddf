import os, sys, torch
sys.path.insert(0, '/root/repo')
from bench import event_time_ms
from omniserve_amd.backend import qgemm_w4a8_per_chn
dev = torch.device("cuda:0")
for (M, N, K) in [(4096, 4096, 4096), (16384, 4096, 14336)]:
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8)
    sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = torch.full((N,), 0.05, dtype=torch.float16, device=dev)
    sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = torch.zeros((M,), dtype=torch.float16, device=dev)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    for iters, warm in ((10, 3), (100, 20), (1000, 200)):
        ms = event_time_ms(lambda i: qgemm_w4a8_per_chn.gemm_forward_cuda(a, w, sw, sa, sz, asum, out), iters=iters, warm=warm)
        print(M, N, K, "iters", iters, "warm", warm, "%.4f ms  %.0f TOPS" % (ms, 2.0 * M * N * K / ms / 1e9), flush=True)
