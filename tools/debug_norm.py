"""Diagnose the rms_norm_general_fuse_sum `sum` mismatch: which 1-ulp perturbation of the oracle's
mean / rstd reproduces the GPU value?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import elementwise as oe
import omniserve_backend.layernorm_ops as ln
F32, F16 = np.float32, np.float16
dev = torch.device("cuda:0")
for tokens, hidden in [(16, 4096), (7, 5120), (33, 8192)]:
    rng = np.random.default_rng(3 * tokens + hidden)
    x = (rng.standard_normal((tokens, hidden)) * 2.0).astype(np.float16) + np.float16(0.25)
    g = (1.0 + 0.1 * np.random.default_rng(1).standard_normal(hidden)).astype(np.float16)
    out = torch.empty((tokens, hidden), dtype=torch.int8, device=dev)
    scale = torch.empty((tokens,), dtype=torch.float16, device=dev)
    ssum = torch.empty((tokens,), dtype=torch.float16, device=dev)
    ln.rms_norm_general_fuse_sum(out, torch.from_numpy(x).to(dev), torch.from_numpy(g).to(dev), ssum, scale, 1e-5, True)
    got = ssum.cpu().numpy()
    xf, gf, nt = x.astype(F32), g.astype(F32), 1024
    psum = oe._thread_partials(xf, nt, lambda a, c: (a + c).astype(F32), 0.0)
    mean0 = (oe.ref_tree_sum(psum) / F32(hidden)).astype(F32)
    pvar = oe._thread_partials((xf * xf).astype(F32), nt, lambda a, c: (a + c).astype(F32), 0.0)
    var = oe.ref_tree_sum(pvar)
    rstd0 = (F32(1.0) / np.sqrt(((var / F32(hidden)).astype(F32) + F32(1e-5)).astype(F32))).astype(F32)

    def fsum(mean, rstd):
        y = ((xf - mean[:, None]).astype(F32) * rstd[:, None]).astype(F32)
        y = (y * gf[None, :]).astype(F32)
        yh = y.astype(F16)
        part = np.zeros((tokens, nt), F16)
        for st in range(0, hidden, nt):
            part = (part.astype(F32) + yh[:, st:st + nt].astype(F32)).astype(F32).astype(F16)
        return oe.ref_tree_sum(part.astype(F32)).astype(F16)

    def bump(a, d):
        b = a.copy(); b.view(np.int32)[:] += d; return b
    base = fsum(mean0, rstd0)
    bad = np.where(got.view(np.uint16) != base.view(np.uint16))[0]
    print("shape", tokens, hidden, "mismatching tokens", bad.tolist())
    for dm in (-1, 0, 1):
        for dr in (-1, 0, 1):
            v = fsum(bump(mean0, dm), bump(rstd0, dr))
            hit = [int(t) for t in bad if v.view(np.uint16)[t] == got.view(np.uint16)[t]]
            print("  mean%+d rstd%+d explains tokens %s" % (dm, dr, hit))
