"""Prefill GEMM A/B across library builds: time + an order-independent checksum of the fp16 output per shape.

    OMNI_TUNE_LIB=tune_libs/libX.so python tools/gemm_ab.py [chn|grp|w8] [--int-mm]

The checksum (sum and xor of the output's 16-bit patterns) must agree between builds: the kernels are bit-exact
re-implementations of each other.  --int-mm also times torch._int_mm (the vendor int8 GEMM, no dequant / epilogue) on the
same shapes as a yardstick for what this chip sustains on int8 at these sizes."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402

if os.environ.get("OMNI_TUNE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["OMNI_TUNE_LIB"])
from bench import event_time_ms  # noqa: E402
from omniserve_amd.backend import qgemm_w4a8_per_chn, qgemm_w4a8_per_group, qgemm_w8a8  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "chn"
dev = torch.device("cuda:0")
SHAPES = [(4096, 4096, 4096), (16384, 6144, 4096), (16384, 28672, 4096), (16384, 4096, 14336), (256, 512, 256), (384, 256, 1024)]
g = torch.Generator(device=dev).manual_seed(1)
for (M, N, K) in SHAPES:
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
    fn(0)
    torch.cuda.synchronize()
    bits = out.view(torch.int16).to(torch.int64) & 0xFFFF
    chk = [int(bits.sum().item()), int((bits * (torch.arange(bits.numel(), device=dev).view_as(bits) % 8191 + 1)).sum().item() % (1 << 61))]
    ms = event_time_ms(fn, iters=10) if M >= 4096 else 0.0
    rec = {"mode": mode, "M": M, "N": N, "K": K, "ms": round(ms, 4), "int8_tops": round(2.0 * M * N * K / ms / 1e9, 1) if ms else None,
           "chk": chk}
    if "--int-mm" in sys.argv and M >= 4096 and mode != "w8":
        wt = torch.randint(-127, 128, (K, N), dtype=torch.int8, device=dev, generator=g)
        ms2 = event_time_ms(lambda i: torch._int_mm(a, wt), iters=10)
        rec["int_mm_tops"] = round(2.0 * M * N * K / ms2 / 1e9, 1)
        del wt
    print(json.dumps(rec), flush=True)
    del a, w, out
