"""Does the static head -> XCD map of the prefill attention lose time when whole kv-head groups are streaming heads (the real
LServe layout: head classes per KV head) instead of alternating q heads (bench.py's synthetic layer)?  One layer, 32 q / 8 kv
heads, sink 128 / local 8192."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from block_sparse_attn import token_streaming_attn_func  # noqa: E402
from omniserve_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
Hq, Hk, D = 32, 8, 128
for L in [int(a) for a in sys.argv[1:]] or [65536]:
    q = torch.randn((L, Hq, D), dtype=torch.float16, device=dev)
    k = torch.randn((L, Hk, D), dtype=torch.float16, device=dev)
    v = torch.randn_like(k)
    cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
    si = torch.tensor([128, 8192] * Hq, dtype=torch.int32, device=dev)
    win = 128 + 8192
    flops = 4.0 * D * (Hq // 2) * (L * L / 2 + (L * L / 2 if L <= win else L * win - win * win / 2))
    pats = {"all 32 q heads dense": [0] * Hq, "all 32 q heads streaming": [-1] * Hq,
            "alternating q heads": [0, -1] * (Hq // 2),
            "alternating kv heads (LServe)": sum(([0] * 4 if g % 2 == 0 else [-1] * 4 for g in range(Hk)), []),
            "first 4 kv heads dense": [0] * 16 + [-1] * 16,
            "3 dense kv heads (0, 3, 5)": sum(([0] * 4 if g in (0, 3, 5) else [-1] * 4 for g in range(Hk)), [])}
    for name, pat, W in [(n, p_, w) for n, p_ in pats.items() for w in (1, 2, 4, 8)]:
        _lib.lib().omni_prefill_set_xcd_split(W)
        hm = torch.tensor(pat, dtype=torch.int32, device=dev)
        nd = sum(1 for t in pat if t == 0)
        fl = flops * nd / 16 + (0 if nd == 16 else 0)
        for _ in range(1):
            token_streaming_attn_func(q, k, v, cu, cu, hm, si, L, L)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(2):
            token_streaming_attn_func(q, k, v, cu, cu, hm, si, L, L)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 2
        print("L=%6d %-32s W=%d %8.2f ms  (%d dense q heads)" % (L, name, W, ms, nd), flush=True)
