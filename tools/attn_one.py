"""Two launches of the prefill attention (one layer, 32 q / 8 kv heads) for PMC passes: python tools/attn_one.py L [dense|mixed]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from block_sparse_attn import flash_attn_varlen_func, token_streaming_attn_func  # noqa: E402

dev = torch.device("cuda:0")
if os.environ.get("OMNI_PREFILL_VARIANT"):
    from omniserve_amd import _lib
    _lib.lib().omni_prefill_set_variant(int(os.environ["OMNI_PREFILL_VARIANT"]))
L = int(sys.argv[1])
mode = sys.argv[2] if len(sys.argv) > 2 else "dense"
Hq, Hk, D = 32, 8, 128
q = torch.randn((L, Hq, D), dtype=torch.float16, device=dev)
k = torch.randn((L, Hk, D), dtype=torch.float16, device=dev)
v = torch.randn_like(k)
cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
for _ in range(int(os.environ.get("OMNI_REPS", "2"))):
    if mode == "dense":
        flash_attn_varlen_func(q, k, v, cu, cu, L, L, causal=True)
    else:
        hm = torch.tensor([0, -1] * (Hq // 2), dtype=torch.int32, device=dev)
        si = torch.tensor([128, 8192] * Hq, dtype=torch.int32, device=dev)
        token_streaming_attn_func(q, k, v, cu, cu, hm, si, L, L)
torch.cuda.synchronize()
