"""16-row vs 32-row form of the prefill attention at long L (one layer, 32 q / 8 kv heads): all dense, and the LServe layout."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402

if os.environ.get("OMNI_TUNE_LIB"):     # A/B against a library variant (tools/build_variant.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["OMNI_TUNE_LIB"])
from block_sparse_attn import flash_attn_varlen_func, token_streaming_attn_func  # noqa: E402

dev = torch.device("cuda:0")
Hq, Hk, D = 32, 8, 128
for L in [int(a) for a in sys.argv[1:]] or [131072]:
    q = torch.randn((L, Hq, D), dtype=torch.float16, device=dev)
    k = torch.randn((L, Hk, D), dtype=torch.float16, device=dev)
    v = torch.randn_like(k)
    cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
    si = torch.tensor([128, 8192] * Hq, dtype=torch.int32, device=dev)
    hm = torch.tensor(sum(([0] * 4 if g % 2 == 0 else [-1] * 4 for g in range(Hk)), []), dtype=torch.int32, device=dev)
    for variant in (0, 1):
        _lib.lib().omni_prefill_set_variant(variant)
        for name, fn in (("all dense", lambda: flash_attn_varlen_func(q, k, v, cu, cu, L, L, causal=True)),
                         ("LServe 4+4", lambda: token_streaming_attn_func(q, k, v, cu, cu, hm, si, L, L))):
            fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                fn()
            b.record()
            torch.cuda.synchronize()
            print("L=%6d %s-row form  %-10s %8.2f ms" % (L, "32" if variant else "16", name, a.elapsed_time(b) / 3), flush=True)
    _lib.lib().omni_prefill_set_variant(0)
