"""Runs the dense 16 K prefill attention (or the 4096^3 W4A8 GEMM, or the decode GEMV) in a loop for N seconds: a load to sample rocm-smi clocks / power against."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib
if os.environ.get('OMNI_TUNE_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['OMNI_TUNE_LIB'])
what = sys.argv[1] if len(sys.argv) > 1 else "attn"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
dev = torch.device("cuda:0")
if what == "attn":
    from flash_attn.flash_attn_interface import flash_attn_varlen_func
    L, Hq, Hk, D = 16384, 32, 8, 128
    q = torch.randn((L, Hq, D), dtype=torch.float16, device=dev); k = torch.randn((L, Hk, D), dtype=torch.float16, device=dev); v = torch.randn_like(k)
    cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
    fn = lambda: flash_attn_varlen_func(q, k, v, cu, cu, L, L, causal=True)
elif what == "gemm":
    from omniserve_amd.backend import qgemm_w4a8_per_chn
    M = N = K = 8192
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8)
    sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = sw.clone()
    sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = sa.clone()
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    fn = lambda: qgemm_w4a8_per_chn.gemm_forward_cuda(a, w, sw, sa, sz, asum, out)
else:
    from omniserve_amd.runtime import DecodeRunner, LlamaConfig
    r = DecodeRunner(LlamaConfig.llama3_8b(-1), 16, 1024, 4000, dev, seed=0, use_graph=True, fused=2)
    fn = r.step
fn(); torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(20):
        fn()
    torch.cuda.synchronize(); n += 20
print("%s: %d iterations in %.2f s = %.3f ms each" % (what, n, time.time() - t0, (time.time() - t0) / n * 1e3))
