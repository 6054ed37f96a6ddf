"""Main-loop wait breakdown of the prefill W4A8 GEMM (debug build, OMNI_HIPCC_EXTRA=-DOMNI_DEBUG_CLOCKS)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib
from omniserve_amd.backend import qgemm_w4a8_per_chn
dev = torch.device("cuda:0")
lib = _lib.lib()
f = lib.omni_debug_clocks_gemm_chn; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 32)()
for (M, N, K) in [(4096, 4096, 4096), (16384, 28672, 4096)]:
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8)
    sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = sw.clone()
    sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = sa.clone()
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    for _ in range(3):
        qgemm_w4a8_per_chn.gemm_forward_cuda(a, w, sw, sa, sz, asum, out)
    torch.cuda.synchronize()
    assert f(buf) == 0
    v = list(buf)
    for wv in range(4):
        print("M=%d N=%d K=%d wave %d: loop %.2f us, barrier wait %.2f us, staged-load wait %.2f us" % (
            M, N, K, wv, v[wv * 4] / 100.0, v[wv * 4 + 1] / 100.0, v[wv * 4 + 2] / 100.0))
