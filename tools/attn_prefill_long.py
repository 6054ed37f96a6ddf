"""Prefill attention alone at a given length (dense + streaming heads as in the LServe leg): python tools/attn_prefill_long.py L"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from block_sparse_attn import token_streaming_attn_func  # noqa: E402

dev = torch.device("cuda:0")
for L in [int(a) for a in sys.argv[1:]] or [16384, 256000]:
    Hq, Hk, D = 32, 8, 128
    q = torch.randn((L, Hq, D), dtype=torch.float16, device=dev)
    k = torch.randn((L, Hk, D), dtype=torch.float16, device=dev)
    v = torch.randn_like(k)
    cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
    hm = torch.tensor([0, -1] * (Hq // 2), dtype=torch.int32, device=dev)
    si = torch.tensor([128, 8192] * Hq, dtype=torch.int32, device=dev)
    token_streaming_attn_func(q, k, v, cu, cu, hm, si, L, L)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    token_streaming_attn_func(q, k, v, cu, cu, hm, si, L, L)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    win = min(L, 128 + 8192)
    flops = 4.0 * D * (Hq // 2) * (L * L / 2 + L * win - (win * win / 2 if L > win else L * L / 2))
    print("L=%d: %.2f ms, %.1f TFLOP/s" % (L, ms, flops / ms * 1e-9))
    del q, k, v
