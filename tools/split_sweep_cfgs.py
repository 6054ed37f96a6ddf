"""Decode step of the secondary configurations against the KV-split count of the dense decode attention (0 = the planner's):
one Llama-2-70B TP = 8 rank at bs = 128, Llama-3-8B g128 at bs = 64."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib
from omniserve_amd.runtime import DecodeRunner, LlamaConfig
dev = torch.device("cuda:0")
lib = _lib.lib()
for name, mk in (("70B TP=8 rank bs=128", lambda: DecodeRunner(LlamaConfig.llama2_70b(-1), 128, 1024, 60, dev, seed=3, tp_rank=0, tp_size=8)),
                 ("8B g128 bs=64", lambda: DecodeRunner(LlamaConfig.llama3_8b(128), 64, 1024, 60, dev, seed=0))):
    for ns in (0, 1, 2, 3, 4):
        lib.omni_kv4_decode_set_split_override(ns)
        r = mk()
        for _ in range(6):
            r.step()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(16):
                r.step()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 16)
        print("%s splits %d: %.4f ms/step" % (name, ns, best * 1e3), flush=True)
        del r
        torch.cuda.empty_cache()
lib.omni_kv4_decode_set_split_override(0)
