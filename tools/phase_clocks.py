"""Phase timing of the latency-bound decode kernels (debug build: OMNI_HIPCC_EXTRA=-DOMNI_DEBUG_CLOCKS).
Runs the bench decode step and prints the 100 MHz wall-clock deltas recorded by workgroup 0 of the last
instance of each kernel."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402

if os.environ.get("OMNI_TUNE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["OMNI_TUNE_LIB"])
from omniserve_amd.runtime import DecodeRunner, LlamaConfig  # noqa: E402

dev = torch.device("cuda:0")
cfg = LlamaConfig.llama3_8b(-1)
r = DecodeRunner(cfg, 16, 1024, 32, dev, seed=1, use_graph=True, fused=int(os.environ.get('OMNI_FUSED', '3')))
for _ in range(6):
    r.step()
torch.cuda.synchronize()
lib = _lib.lib()
buf = (ctypes.c_ulonglong * 32)()


def show(fn, names, base):
    f = getattr(lib, fn)
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p]
    assert f(buf) == 0
    v = list(buf)
    t0 = v[base]
    for k, nm in names:
        print("  %-34s +%.2f us" % (nm, (v[k] - t0) / 100.0))


print("general_norm_v2 (slab + add + norm + quant + sum):")
show("omni_debug_clocks_elementwise", [(0, "entry"), (1, "inputs loaded, xs written"), (2, "ordered partials done"),
                                        (3, "tree_sum8<2> done"), (4, "normalised, y in xs"), (5, "block max done"),
                                        (6, "int8 codes stored"), (7, "fp16 sum replayed, stored")], 0)
print("quant_v2 (last instance: attention merge or silu):")
show("omni_debug_clocks_elementwise", [(8, "entry"), (9, "batched fetch done"), (10, "values computed"),
                                        (11, "block max done"), (12, "sum tree done"), (13, "int8 stored")], 8)
print("kv4_decode_flash_kernel (workgroup 0; every mark drains the memory counters first):")
show("omni_debug_clocks_kv", [(16, "entry"), (17, "trip 1 done (len, pages, q/k/v, rope coefs)"),
                              (18, "first K/V batch landed, RoPE(q) in LDS"), (19, "tile sweep done"),
                              (22, "combine + partials written")], 16)
