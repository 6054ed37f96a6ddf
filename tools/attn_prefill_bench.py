"""Prefill attention throughput, dense causal (the 16 K / 32-head figure of DESIGN 4.5) and the LServe mix at 64 K.
    [OMNI_TUNE_LIB=tune_libs/x.so] python tools/attn_prefill_bench.py
Checks the first case against a float64 evaluation of a few query rows (a variant build must still be right)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402

if os.environ.get("OMNI_TUNE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["OMNI_TUNE_LIB"])
from block_sparse_attn import token_streaming_attn_func  # noqa: E402
from flash_attn.flash_attn_interface import flash_attn_varlen_func  # noqa: E402

dev = torch.device("cuda:0")
Hq, Hk, D = 32, 8, 128
if os.environ.get("OMNI_PREFILL_VARIANT"):      # 0: 16-row form, 1: 32-row form, 2: ping-pong schedule of the 16-row form
    _lib.lib().omni_prefill_set_variant(int(os.environ["OMNI_PREFILL_VARIANT"]))
    print("variant", os.environ["OMNI_PREFILL_VARIANT"])


def run(L, mixed, reps=3):
    g = torch.Generator(device=dev).manual_seed(L)
    q = torch.randn((L, Hq, D), dtype=torch.float16, device=dev, generator=g)
    k = torch.randn((L, Hk, D), dtype=torch.float16, device=dev, generator=g)
    v = torch.randn((L, Hk, D), dtype=torch.float16, device=dev, generator=g)
    cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
    hm = torch.tensor([0, -1] * (Hq // 2), dtype=torch.int32, device=dev)
    si = torch.tensor([128, 8192] * Hq, dtype=torch.int32, device=dev)
    fn = (lambda: token_streaming_attn_func(q, k, v, cu, cu, hm, si, L, L)) if mixed else \
         (lambda: flash_attn_varlen_func(q, k, v, cu, cu, L, L, causal=True))
    out = fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    if mixed:
        win = min(L, 128 + 8192)
        flops = 4.0 * D * (Hq // 2) * (L * L / 2 + L * win - (win * win / 2 if L > win else L * L / 2))
    else:
        flops = 4.0 * D * Hq * L * L / 2
    # spot check: 3 rows x 2 heads in float64
    err = 0.0
    for row in (L // 3, L - 1, 70):
        for h in (0, Hq - 1):
            if mixed and h % 2 == 1:
                continue
            kk = k[: row + 1, h // (Hq // Hk)].double()
            s_ = (kk @ q[row, h].double()) / np.sqrt(D)
            p_ = torch.softmax(s_, 0)
            ref = p_ @ v[: row + 1, h // (Hq // Hk)].double()
            err = max(err, float((out[row, h].double() - ref).abs().max() / ref.abs().max()))
    print("L=%6d %s: %8.3f ms  %7.1f TFLOP/s  (spot-check rel err %.2e)" % (L, "mixed" if mixed else "dense", best, flops / best * 1e-9, err), flush=True)


run(16384, False)
run(4096, False)
run(65536, True)
