"""W8A8 GEMM at prefill shapes (Llama-3-8B, M = 16384) and 4096^3."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402

if os.environ.get("OMNI_TUNE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["OMNI_TUNE_LIB"])
from bench import event_time_ms  # noqa: E402
from omniserve_amd.backend import qgemm_w8a8  # noqa: E402

dev = torch.device("cuda:0")
for (M, N, K) in [(4096, 4096, 4096), (16384, 6144, 4096), (16384, 28672, 4096), (16384, 4096, 14336), (1000, 4096, 4096)]:
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev)
    sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev)
    sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    ms = event_time_ms(lambda i: qgemm_w8a8.w8a8_gemm_forward_cuda(a, w, sw, sa, out), iters=10)
    print(json.dumps({"M": M, "N": N, "K": K, "ms": round(ms, 4), "int8_tops": round(2.0 * M * N * K / ms / 1e9, 1)}), flush=True)
    del a, w, out
