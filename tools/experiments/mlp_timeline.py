#!/usr/bin/env python
"""Per-phase timeline of the persistent MLP launch (csrc/mlp_fused.hip) and its A/B against the three-launch sequence it
replaces, at the BASELINE configs[1] layer (M = 16, hidden 4096, inter 14336), weights rotated over 32 layers (cold).

    python tools/mlp_timeline.py [--layers 32] [--inter 14336] [--m 16]

Prints: us per layer of (a) splitk_add_rms_norm_general_fuse_sum [+ L2 prefetch of down_proj riding on it] -> gemm_silu ->
gemm_partial_f16 in one HIP graph, (b) the fused launch in one HIP graph; and, from the launch's own 100 MHz wall-clock marks
(one per workgroup and phase), when each phase starts / ends across the 256 workgroups."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd.backend import fused_ext  # noqa: E402

MARKS = ["start", "norm row published (service)", "hand-off 1 passed", "first gate_up unit done", "arrived at hand-off 2",
         "hand-off 2 passed", "down units done", "end (rider done)"]


def graph_time(fn, reps=8):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps, g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--inter", type=int, default=14336)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--prefetch-mb", type=float, default=40.0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    M, H, I, NL = args.m, 4096, args.inter, args.layers
    g = torch.Generator(device=dev); g.manual_seed(1)
    rnd = lambda *s: torch.rand(s, device=dev, generator=g)  # noqa: E731
    L = []
    for _ in range(NL):
        L.append(dict(Wgu=torch.randint(0, 256, (2 * I, H // 2), dtype=torch.uint8, device=dev, generator=g).view(torch.int8),
                      gu_ws=(0.002 + 0.018 * rnd(2 * I)).half(), gu_wsz=(0.05 * rnd(2 * I)).half(),
                      Wdn=torch.randint(0, 256, (H, I // 2), dtype=torch.uint8, device=dev, generator=g).view(torch.int8),
                      gamma=(1.0 + 0.05 * torch.randn(H, device=dev, generator=g)).half(),
                      o_ws=(0.002 + 0.004 * rnd(H)).half(), o_wsz=(0.02 * rnd(H)).half()))
    x0 = (0.7 * torch.randn((M, H), device=dev, generator=g)).half()
    o_slab = torch.randint(-40000, 40000, (3, M, H), dtype=torch.int32, device=dev, generator=g)
    o_as = (0.005 + 0.01 * rnd(M)).half(); o_asum = (3.0 * torch.randn(M, device=dev, generator=g)).half()
    x = x0.clone()
    q = torch.empty((M, H), dtype=torch.int8, device=dev)
    sB = torch.empty((M,), dtype=torch.float16, device=dev); mB = torch.empty_like(sB)
    s2 = torch.empty_like(sB); m2 = torch.empty_like(sB)
    act = torch.empty((M, I), dtype=torch.float16, device=dev)
    amax = torch.zeros((NL, fused_ext.AMAX_WORDS), dtype=torch.int32, device=dev)
    slab = torch.empty((8, M, H), dtype=torch.int32, device=dev)
    counters, scratch = fused_ext.mlp_fused_buffers(NL, H, I, dev)
    pf = int(args.prefetch_mb * (1 << 20))

    def three(prefetch):
        amax.zero_()
        for li, P in enumerate(L):
            if prefetch:
                fused_ext.prefetch_arm_gemm(P["Wdn"], M, H, I, 0, True, pf, 160)
            fused_ext.splitk_add_rms_norm_general_fuse_sum(q, x, o_slab, 3, P["o_ws"], o_as, P["o_wsz"], o_asum, P["gamma"], mB, sB, 1e-5)
            fused_ext.gemm_silu_per_chn(q, P["Wgu"], P["gu_ws"], sB, P["gu_wsz"], mB, act, amax[li])
            fused_ext.gemm_partial_f16_per_chn(act, amax[li], P["Wdn"], slab, m2, s2)

    def fused(clocks=False):
        counters.zero_()
        for li, P in enumerate(L):
            fused_ext.mlp_fused_per_chn(x, o_slab, 3, P["o_ws"], P["o_wsz"], o_as, o_asum, P["gamma"], 1e-5, P["Wgu"], P["gu_ws"],
                                        P["gu_wsz"], P["Wdn"], slab, m2, s2, counters, NL, li, scratch, clocks=clocks)

    for name, fn in (("three launches, no prefetch", lambda: three(False)),
                     ("three launches, down_proj prefetched by the norm (level 3 as it runs in the step)", lambda: three(True)),
                     ("fused persistent launch", lambda: fused(False))):
        x.copy_(x0)
        ms, _ = graph_time(fn)
        print("%-90s %8.2f us per layer" % (name, ms * 1e3 / NL))
    fused_ext.mlp_fused_check(counters)

    # ---- timeline: one eager launch with clocks (the last layer's marks survive in the scratch tail)
    x.copy_(x0)
    fused(clocks=True)
    torch.cuda.synchronize()
    fused_ext.mlp_fused_check(counters)
    tail = scratch[H * 16 + 256 + I * 32:].view(torch.int64).view(256, 8).cpu().numpy().astype(np.float64) * 0.01   # us
    t0 = tail[:, 0].min()
    rel = tail - t0
    heavy = np.arange(256) >= 64
    print("\nphase marks of the last launch, us after the first workgroup's start (min / median / max over workgroups):")
    for k, name in enumerate(MARKS):
        col = rel[:, k]
        sel = col[col > -1e6]
        if k == 1:
            sel = rel[:16, 1]
        if k == 3:
            print("  %-36s heavy (2 + 2 units): %6.2f / %6.2f / %6.2f   light: %6.2f / %6.2f / %6.2f" % (
                name, rel[heavy, k].min(), np.median(rel[heavy, k]), rel[heavy, k].max(),
                rel[~heavy, k].min(), np.median(rel[~heavy, k]), rel[~heavy, k].max()))
            continue
        print("  %-36s %6.2f / %6.2f / %6.2f" % (name, sel.min(), np.median(sel), sel.max()))
    print("  service workgroups (0..15) end at   %6.2f .. %6.2f" % (rel[:16, 7].min(), rel[:16, 7].max()))


if __name__ == "__main__":
    main()
