"""Decode step against the form of the dense KV4 decode sweep (omni_kv4_decode_set_raw_override: 0 exact fp16 dequantisation,
1 raw codes on the matrix cores), alternating on one box: python tools/attn_raw_ab.py [model] [batch ...]
Needs a library built with tools/experiments/attention_raw_code_sweep.patch applied (the shipped one has no such override)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402

if os.environ.get("OMNI_TUNE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["OMNI_TUNE_LIB"])
    _lib.USE_EXT = False
from omniserve_amd.runtime import DecodeRunner, LlamaConfig  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib()
model = sys.argv[1] if len(sys.argv) > 1 else "8b"
batches = [int(x) for x in sys.argv[2:]] or [16, 64, 128, 256]
cfg = {"8b": lambda: LlamaConfig.llama3_8b(-1), "8b_g128": lambda: LlamaConfig.llama3_8b(128),
       "70b": lambda: LlamaConfig.llama2_70b(-1)}[model]()
ctx = int(os.environ.get("CTX", "1024"))
for batch in batches:
    res = {0: [], 1: []}
    for rep in range(2):
        for raw in (0, 1):
            lib.omni_kv4_decode_set_raw_override(raw)
            r = DecodeRunner(cfg, batch, ctx, 200, dev, seed=0, use_graph=True)
            for _ in range(6):
                r.step()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(24):
                    r.step()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 24)
            res[raw].append(best * 1e3)
            del r
            torch.cuda.empty_cache()
    print("%s bs %3d ctx %d: exact %s  raw %s ms/step" % (model, batch, ctx, " ".join("%.4f" % x for x in res[0]),
                                                         " ".join("%.4f" % x for x in res[1])), flush=True)
lib.omni_kv4_decode_set_raw_override(-1)
