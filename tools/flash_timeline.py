"""Per-wave shader-clock timeline of kv4_decode_flash_kernel for the first / middle / last workgroup of the grid (debug build:
tools/build_variant.sh flash_clk "-DOMNI_DEBUG_CLOCKS"; OMNI_TUNE_LIB=tune_libs/libflash_clk.so python tools/flash_timeline.py B T)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kernel_bench import D, Pools, dev  # noqa: E402  (honours OMNI_TUNE_LIB)
from omniserve_amd import _lib  # noqa: E402
import omniserve_backend.fused_attention_pure_dense as pd  # noqa: E402

B, Tc = int(sys.argv[1]), int(sys.argv[2])
single = len(sys.argv) > 3 and sys.argv[3] == "single"      # the single-launch form of fusion level 3 (last arriver merges)
Hq, Hk = 32, 8
pools = Pools(B, Tc // 64 + 2, Hk, row=64)
lens = torch.full((B,), Tc + 1, dtype=torch.int32, device=dev)
q = torch.randn((B, Hq, D), dtype=torch.float16, device=dev)
k = torch.randn((B, Hk, D), dtype=torch.float16, device=dev)
v = torch.randn_like(k)
if single:
    from omniserve_amd.backend import fused_ext
    out = torch.empty((B, Hq * D), dtype=torch.float16, device=dev)
    amax = fused_ext.new_amax_slots(B, dev)
for _ in range(3):
    if single:
        fused_ext.decode_attention_f16_amax(out, amax, q, k, v, pools.table, lens, 64, Tc + 1, 500000.0)
    else:
        pd.single_query_attention(q, k, v, pools.table, lens, None, 1 << 20, 64, Hk * D // 2, Tc + 1, D, 500000.0, True, True, True)
torch.cuda.synchronize()
f = _lib.lib().omni_debug_timeline_flash
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p]
buf = np.zeros((3, 4, 32), dtype=np.uint64)
assert f(buf.ctypes.data) == 0
t = buf.astype(np.int64)
base = t[0, :, 0].min()
for w, name in enumerate(("first", "middle", "last")):
    print("== %s workgroup (cycles; columns = waves 0..3)" % name)
    e = t[w, :, 0]
    print("entry (from the first workgroup's)  ", " ".join("%7d" % (x - base) for x in e))
    print("page window visible   +", " ".join("%7d" % x for x in (t[w, :, 1] - e)))
    print("q in LDS              +", " ".join("%7d" % x for x in (t[w, :, 2] - t[w, :, 1])))
    prev = t[w, :, 2]
    for i in range(22):
        cur = t[w, :, 3 + i]
        if (cur == 0).all():
            break
        print("tile %2d                +" % i, " ".join("%7d" % (c - p_) if c else "      -" for c, p_ in zip(cur, prev)))
        prev = np.where(cur > 0, cur, prev)
    print("sweep done            +", " ".join("%7d" % x for x in (t[w, :, 28] - prev)))
    if single:
        print("partials stored       +", " ".join("%7d" % x for x in (t[w, :, 25] - t[w, :, 28])))
        print("stores drained        +", " ".join("%7d" % x for x in (t[w, :, 26] - t[w, :, 25])))
        print("ticket known          +", " ".join("%7d" % x for x in (t[w, :, 27] - t[w, :, 26])))
        print("merged (last arriver) +", " ".join("%7d" % x for x in (t[w, :, 29] - t[w, :, 27])))
    else:
        print("combined / stored     +", " ".join("%7d" % x for x in (t[w, :, 29] - t[w, :, 28])))
    print("total                  ", " ".join("%7d" % x for x in (t[w, :, 29] - e)))
