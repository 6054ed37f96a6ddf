"""Decode step time of configs[1] (per-channel, bs 16) and configs[2] (g128, bs 64) for A/B runs of a library variant
(OMNI_TUNE_LIB=path): python tools/step_ab.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402

if os.environ.get("OMNI_TUNE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["OMNI_TUNE_LIB"])
from omniserve_amd.runtime import DecodeRunner, LlamaConfig  # noqa: E402

dev = torch.device("cuda:0")
if os.environ.get("OMNI_NSPLIT"):      # KV split override of the decode attention (0 = the planner's choice)
    _lib.lib().omni_kv4_decode_set_split_override(int(os.environ["OMNI_NSPLIT"]))
for gs, bs in ((-1, 16), (128, 64)):
    r = DecodeRunner(LlamaConfig.llama3_8b(gs), bs, 1024, 200, dev, seed=0, fused=int(os.environ.get('OMNI_FUSED', '3' if bs <= 16 else '2')))
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
    print("group %4d bs %3d: %.4f ms/step  %.1f tok/s  tokens %s" % (gs, bs, best * 1e3, bs / best, r.tokens[:4].tolist()), flush=True)
    del r
    torch.cuda.empty_cache()
