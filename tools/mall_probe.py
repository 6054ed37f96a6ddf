"""Is a decode GEMV faster when its weights are resident in the 256 MB Infinity Cache (MALL)?
Times the gate_up / down GEMV (graph-timed) over 1, 2 and 12 rotating weight copies, and with a
concurrent prefetch (plain read) of the next copy on a second stream."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd.backend import qgemm_w4a8_per_chn
from tools.sweep_graph import graph_time_us
dev = torch.device("cuda:0")
M = 16
for (N, K) in [(28672, 4096), (4096, 14336), (4096, 4096)]:
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = sw.clone()
    sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = sa.clone()
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    alg = N * K // 2
    for copies in (1, 2, 3, 4, 12):
        ws = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8) for _ in range(copies)]
        us = graph_time_us(lambda i: qgemm_w4a8_per_chn.gemm_forward_cuda(a, ws[i % copies], sw, sa, sz, asum, out), max(copies, 4))
        print("N=%d K=%d copies=%2d (%6.1f MB working set): %7.2f us %7.1f GB/s" % (N, K, copies, copies * alg / 1e6, us, alg / us / 1e3), flush=True)
        del ws

# prefetch experiment: touch copy i+1 with a plain torch reduction on a side stream while GEMV i runs
N, K = 28672, 4096
copies = 12
ws = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8) for _ in range(copies)]
a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = sw.clone()
sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = sa.clone()
out = torch.empty((M, N), dtype=torch.float16, device=dev)
side = torch.cuda.Stream()
sink = torch.zeros((1,), dtype=torch.int32, device=dev)
def step_serial(i):
    _ = ws[(i + 1) % copies].view(torch.int32).sum()          # "prefetch" as a plain read, same stream (serial)
    qgemm_w4a8_per_chn.gemm_forward_cuda(a, ws[i % copies], sw, sa, sz, asum, out)
us = graph_time_us(step_serial, copies)
print("gate_up GEMV + serial read of the next copy: %7.2f us per pair" % us, flush=True)
def step_gemv_after_touch(i):
    _ = ws[i % copies].view(torch.int32).sum()                 # touch THIS copy first (warms MALL), then GEMV on it
    qgemm_w4a8_per_chn.gemm_forward_cuda(a, ws[i % copies], sw, sa, sz, asum, out)
us2 = graph_time_us(step_gemv_after_touch, copies)
def step_touch_only(i):
    _ = ws[i % copies].view(torch.int32).sum()
us3 = graph_time_us(step_touch_only, copies)
print("touch(copy i) + GEMV(copy i): %7.2f us ; touch alone %7.2f us -> GEMV on MALL-warm weights ~ %7.2f us" % (us2, us3, us2 - us3), flush=True)
