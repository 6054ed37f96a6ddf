"""Decode step (configs[1]) against the KV-split count of the decode attention (planner default vs overrides)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib
from omniserve_amd.runtime import DecodeRunner, LlamaConfig
dev = torch.device("cuda:0")
lib = _lib.lib()
for ns in (0, 2, 3, 4, 5, 6, 8):
    lib.omni_kv4_decode_set_split_override(ns)
    r = DecodeRunner(LlamaConfig.llama3_8b(-1), 16, 1024, 200, dev, seed=0, use_graph=True, fused=3)
    for _ in range(8):
        r.step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(48):
            r.step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 48)
    print("splits %d: %.4f ms/step" % (ns, best * 1e3), flush=True)
    del r
    torch.cuda.empty_cache()
lib.omni_kv4_decode_set_split_override(0)
