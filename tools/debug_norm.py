"""Compare the norm statistics computed on the GPU with the oracle's, stage by stage."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import elementwise as oe
from omniserve_amd import _lib
F32 = np.float32
dev = torch.device("cuda:0")
h = ctypes.CDLL(_lib.LIB_PATH)
h.omni_debug_norm_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
names = ["tot", "var", "mean", "vh", "ve", "sq", "rstd", "lsum_t0"]
for tokens, hidden in [(16, 4096), (7, 5120), (33, 8192)]:
    rng = np.random.default_rng(3 * tokens + hidden)
    x = (rng.standard_normal((tokens, hidden)) * 2.0).astype(np.float16) + np.float16(0.25)
    out = torch.zeros((tokens, 8), dtype=torch.float32, device=dev)
    h.omni_debug_norm_stats(torch.from_numpy(x).to(dev).data_ptr(), out.data_ptr(), 1e-5, tokens, hidden, None)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    xf, nt = x.astype(F32), 1024
    psum = oe._thread_partials(xf, nt, lambda a, c: (a + c).astype(F32), 0.0)
    tot = oe.ref_tree_sum(psum)
    pvar = oe._thread_partials((xf * xf).astype(F32), nt, lambda a, c: (a + c).astype(F32), 0.0)
    var = oe.ref_tree_sum(pvar)
    mean = (tot / F32(hidden)).astype(F32)
    vh = (var / F32(hidden)).astype(F32)
    ve = (vh + F32(1e-5)).astype(F32)
    sq = np.sqrt(ve).astype(F32)
    rstd = (F32(1.0) / sq).astype(F32)
    want = np.stack([tot, var, mean, vh, ve, sq, rstd, psum[:, 0]], axis=1)
    print("shape", tokens, hidden)
    for c, n in enumerate(names):
        bad = np.where(got[:, c].view(np.uint32) != want[:, c].view(np.uint32))[0]
        d = (got[:, c].view(np.int32).astype(np.int64) - want[:, c].view(np.int32).astype(np.int64))
        print("  %-8s mismatches %s ulp-diffs %s" % (n, bad.tolist(), d[bad].tolist()))
