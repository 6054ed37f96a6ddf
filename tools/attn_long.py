"""Long-context KV4 / KV8 decode attention in isolation (B=8, T=32768, GQA 32/8) for rocprofv3 PMC passes.
Usage: python tools/attn_long.py [kv4|kv8] [iters] [B] [T]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kernel_bench import D, Pools, dev, timed  # noqa: E402
import omniserve_backend.fused_attention_per_tensor_dense as ptd  # noqa: E402
import omniserve_backend.fused_attention_pure_dense as pd  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "kv4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
Hq, Hk = int(os.environ.get("OMNI_HQ", 32)), int(os.environ.get("OMNI_HK", 8))      # (one TP = 8 rank of Llama-2-70B: 8 / 1)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
Tc = int(sys.argv[4]) if len(sys.argv) > 4 else 32768
row = 128 if mode == "kv8" else 64
pools = Pools(B, Tc // 64 + 2, Hk, row=row)
lens = torch.full((B,), Tc + 1, dtype=torch.int32, device=dev)
q = torch.randn((B, Hq, D), dtype=torch.float16, device=dev)
k = torch.randn((B, Hk, D), dtype=torch.float16, device=dev)
v = torch.randn_like(k)
if mode == "kv8":
    qo = torch.tensor([0.03, 0.035], dtype=torch.float32, device=dev)
    oq = 1.0 / qo
    flags = torch.ones((Hk,), dtype=torch.int32, device=dev)
    rank = torch.arange(Hk, dtype=torch.int32, device=dev)
    fn = lambda: ptd.single_query_attention(q, k, v, qo, oq, pools.table, None, flags, rank, lens, None, 1 << 20, 64,
                                            Hk * D, 0, 0, 0, 0, 0, Hk, 0, Tc + 1, D, 500000.0, 1.0, True, False, False,
                                            2048)
    nbytes = 2 * Hk * D * Tc * B
else:
    fn = lambda: pd.single_query_attention(q, k, v, pools.table, lens, None, 1 << 20, 64, Hk * D // 2, Tc + 1, D,
                                           500000.0, True, True, True)
    nbytes = 1088 * Tc * B
if os.environ.get("OMNI_NSPLIT"):      # planner override (tuning): KV splits per (sequence, head group)
    from omniserve_amd import _lib as _l
    _l.lib().omni_kv4_decode_set_split_override(int(os.environ["OMNI_NSPLIT"]))
if os.environ.get("OMNI_DBG_POOL"):      # -DOMNI_FLASH_ABLATE=64 builds: trip 1 without its loads (page pointers by arithmetic)
    import ctypes
    from omniserve_amd import _lib as _l
    h = ctypes.CDLL(_l.LIB_PATH)
    h.omni_debug_set_pool.argtypes = [ctypes.c_ulonglong] * 4
    assert h.omni_debug_set_pool(pools.k.data_ptr(), pools.v.data_ptr(), pools.page_bytes, Tc) == 0
us = timed(fn, iters=iters)
print("%s decode attention B=%d T=%d: %.1f us, %.0f GB/s" % (mode, B, Tc, us, nbytes / us / 1e3))
