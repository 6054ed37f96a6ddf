"""Per-phase shader-clock timeline of w4a8_midm_kernel (a -DOMNI_DEBUG_CLOCKS variant: tools/build_variant.sh midm_clk
"-DOMNI_DEBUG_CLOCKS"; OMNI_TUNE_LIB=tune_libs/libmidm_clk.so python tools/midm_timeline.py M N K)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(os.environ["OMNI_TUNE_LIB"])
from omniserve_amd.backend import qgemm_w4a8_per_chn  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4])
dev = torch.device("cuda:0")
copies = max(2, min(16, int(600e6 // (N * K // 2))))
ws = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8) for _ in range(copies)]
a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = sw.clone()
sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = sa.clone()
out = torch.empty((M, N), dtype=torch.float16, device=dev)
lib = _lib.lib()
lib.omni_gemm_set_midm_override(1, 0)
for i in range(3 * copies):
    qgemm_w4a8_per_chn.gemm_forward_cuda(a, ws[i % copies], sw, sa, sz, asum, out)
torch.cuda.synchronize()
f = lib.omni_debug_timeline_midm_chn
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p]
buf = np.zeros((2, 8, 104), dtype=np.uint64)
assert f(buf.ctypes.data) == 0
nch = min(K // 256, 24)
for wg in range(2):
    t = buf[wg].astype(np.int64)
    t0 = t[:, 0].min()
    print("== workgroup %s (shader-clock cycles from the first wave's entry; columns = waves 0..7)" % ("0" if wg == 0 else "last"))
    print("entry      ", " ".join("%6d" % (x - t0) for x in t[:, 0]))
    print("issued D   ", " ".join("%6d" % (x - t0) for x in t[:, 1]))
    for c in range(nch):
        r = t[:, 4 + 4 * c: 8 + 4 * c]
        print("chunk %2d: top %6d | dma wait %s | barrier %s | body %s" % (
            c, r[0, 0] - t0, " ".join("%4d" % x for x in (r[:, 1] - r[:, 0])), " ".join("%4d" % x for x in (r[:, 2] - r[:, 1])),
            " ".join("%4d" % x for x in (r[:, 3] - r[:, 2]))))
    print("K loop done", " ".join("%6d" % (x - t0) for x in t[:, 100]))
    print("exchanged  ", " ".join("%6d" % (x - t0) for x in t[:, 101]))
    print("stored     ", " ".join("%6d" % (x - t0) for x in t[:, 102]))

