"""Step time of the bench decode config vs the KV split count of the decode attention (override hook)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib
from omniserve_amd.runtime import DecodeRunner, LlamaConfig
dev = torch.device("cuda:0")
lib = _lib.lib()
cfg = LlamaConfig.llama3_8b(-1)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for ns in (0, 2, 3, 4, 5, 6, 8, 12, 16):
    lib.omni_kv4_decode_set_split_override(ns)
    r = DecodeRunner(cfg, batch, 1024, 80, dev, seed=1, use_graph=True, fused=2)
    for _ in range(8):
        r.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(48):
        r.step()
    torch.cuda.synchronize()
    print("nsplit override %2d : %.4f ms/step" % (ns, (time.perf_counter() - t0) / 48 * 1e3), flush=True)
    del r
    torch.cuda.empty_cache()
lib.omni_kv4_decode_set_split_override(0)
