"""Round 6: the (norm -> GEMV) pairs as single launches (fusion level 4) against level 3.
    python tools/pairs_ab.py [--steps 48] [--timeline] [--isolated]
  * step A/B (BASELINE configs[1]): ms per decode step at level 3 and level 4 in ONE process, tokens must agree;
  * --isolated: HIP-graph timing of one pair, weights rotated over > MALL copies: two launches (row kernel, GEMV) vs one;
  * --timeline: wall-clock marks of one fused launch (rows: start / body done / published; tiles: start / ring requested /
    gate passed / K loop done / end), min / median / max over the workgroups, us after the first workgroup's start."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd.backend import fused_ext  # noqa: E402
from omniserve_amd.runtime import DecodeRunner, LlamaConfig, W4A8Linear  # noqa: E402
from omniserve_amd import _lib  # noqa: E402


def step_ab(args, dev):
    cfg = LlamaConfig.llama3_8b(args.group_size)
    out = {}
    for level in [int(x) for x in args.levels.split(",")]:
        r = DecodeRunner(cfg, args.batch, args.context, 4 * args.steps + 16, dev, seed=1234, fused=level)
        for _ in range(6):
            r.step()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                r.step()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / args.steps)
        r.check()
        out[level] = (best * 1e3, r.tokens.clone(), r.pairs, r.fused)
        print("level %d (runs as %d, pairs %s): %.4f ms/step  %.1f tok/s" % (level, r.fused, r.pairs, best * 1e3, args.batch / best),
              flush=True)
        del r
        torch.cuda.empty_cache()
    lv = sorted(out)
    for l in lv[1:]:
        print("tokens level %d vs level %d: %s   x%.3f" % (l, lv[0], "same" if torch.equal(out[l][1], out[lv[0]][1]) else "DIFFER",
                                                           out[lv[0]][0] / out[l][0]), flush=True)


class Pair:
    """One (rows from o_proj-like slabs) -> (qkv | gate_up) pair on `copies` rotating weight sets."""

    def __init__(self, dev, M, N, H, silu, copies, seed=5):
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        self.M, self.N, self.H, self.silu = M, N, H, silu
        self.lins = [W4A8Linear(N, H, -1, gen, dev) for _ in range(copies)]
        self.prod = W4A8Linear(H, H, -1, gen, dev)
        a = torch.randint(-127, 128, (M, H), dtype=torch.int8, device=dev, generator=gen)
        self.slab = torch.empty((max(int(_lib.lib().omni_gemm_partial_workspace_bytes(M, H, H)), 1 << 20),), dtype=torch.uint8, device=dev)
        self.sk = fused_ext.gemm_partial_per_chn(a, self.prod.qweight, self.slab)
        f16 = torch.float16
        self.p_sa = (torch.rand((M,), device=dev, generator=gen) * 0.01 + 0.001).to(f16)
        self.p_as = torch.randn((M,), device=dev, generator=gen).to(f16)
        self.gamma = (1.0 + 0.05 * torch.randn((H,), device=dev, generator=gen)).to(f16)
        self.res = torch.randn((M, H), device=dev, generator=gen).to(f16)
        self.codes = torch.empty((M, H), dtype=torch.int8, device=dev)
        self.sum, self.scale = torch.empty((M,), dtype=f16, device=dev), torch.empty((M,), dtype=f16, device=dev)
        self.out = torch.empty((M, N // 2 if silu else N), dtype=f16, device=dev)
        self.amax = fused_ext.new_amax_slots(M, dev) if silu else None
        self.sync = torch.zeros((copies, fused_ext.NGF_SYNC_WORDS), dtype=torch.int32, device=dev)
        self.err = torch.zeros((4,), dtype=torch.int32, device=dev)

    def fused(self, i, clk=None):
        fused_ext.norm_gemm_fused(self.codes, self.res, self.gamma, self.sum, self.scale, 1e-5, self.lins[i], self.out, self.sync[i],
                                  self.err, slab=self.slab, sk=self.sk, producer=self.prod, p_ascales=self.p_sa, p_asums=self.p_as,
                                  amax=self.amax, clk=clk)

    def two(self, i):
        import omniserve_backend.qgemm_w4a8_per_chn as gemm
        L = self.lins[i]
        fused_ext.splitk_add_rms_norm_general_fuse_sum(self.codes, self.res, self.slab, self.sk, self.prod.s1_scales, self.p_sa,
                                                       self.prod.s1_szeros, self.p_as, self.gamma, self.sum, self.scale, 1e-5)
        if self.silu:
            fused_ext.gemm_silu_per_chn(self.codes, L.qweight, L.s1_scales, self.scale, L.s1_szeros, self.sum, self.out, self.amax)
        else:
            gemm.gemm_forward_cuda(self.codes, L.qweight, L.s1_scales, self.scale, L.s1_szeros, self.sum, self.out)


def graph_time(fn, copies, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(copies):
            fn(i)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(copies):
            fn(i)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * copies)


def isolated(args, dev):
    for name, N, silu, copies in (("gate_up", 28672, True, 12), ("qkv", 6144, False, 40)):
        p = Pair(dev, args.batch, N, 4096, silu, copies)

        def fused(i):
            p.sync[i].zero_()     # (a memset node per launch; the decode step zeroes all of a step's words in one kernel)
            p.fused(i)

        def zero_only(i):
            p.sync[i].zero_()

        t2 = graph_time(p.two, copies)
        t1 = graph_time(fused, copies)
        tz = graph_time(zero_only, copies)
        mb = N * 4096 / 2 / 1e6
        print("%-8s two launches %.2f us   one launch %.2f us (incl. %.2f us zeroing node)   weights %.1f MB -> %.2f TB/s vs %.2f TB/s"
              % (name, t2, t1, tz, mb, mb / t2, mb / (t1 - tz)), flush=True)
        assert int(p.err[0].item()) == 0


def timeline(args, dev):
    for name, N, silu, copies in (("gate_up", 28672, True, 12), ("qkv", 6144, False, 40)):
        p = Pair(dev, args.batch, N, 4096, silu, copies)
        grid = args.batch + N // 64
        clk = torch.zeros((grid, 8), dtype=torch.int64, device=dev)
        for i in range(copies):
            p.fused(i)       # rotate through the copies so that the probed launch streams from HBM
        p.sync.zero_()
        torch.cuda.synchronize()
        p.fused(0, clk=clk)
        torch.cuda.synchronize()
        c = clk.cpu().numpy().astype(np.float64) / 100.0      # 100 MHz -> us
        t0 = c[:, 0].min()
        rows, tiles = c[:args.batch] - t0, c[args.batch:] - t0
        q = lambda v: "%6.2f /%6.2f /%6.2f" % (v.min(), np.median(v), v.max())  # noqa: E731
        print("%s  (grid %d; us after the first workgroup's start; min / median / max)" % (name, grid))
        for k, what in ((0, "rows  start"), (5, "rows  inputs loaded"), (1, "rows  body done"), (2, "rows  published")):
            print("   %-22s %s" % (what, q(rows[:, k])))
        for k, what in ((0, "tiles start"), (1, "tiles ring requested"), (2, "tiles gate passed"), (3, "tiles K loop done"), (4, "tiles end")):
            print("   %-22s %s" % (what, q(tiles[:, k])))
        sys.stdout.flush()


def step_timeline(args, dev):
    """The same marks from INSIDE the captured decode step (layer 5's two pair launches)."""
    cfg = LlamaConfig.llama3_8b(args.group_size)
    r = DecodeRunner(cfg, args.batch, args.context, 64, dev, seed=1234, fused=4)
    qkv_n = (cfg.heads + 2 * cfg.kv_heads) * cfg.head_dim
    r.ngf_clk = {(5, 0): torch.zeros((args.batch + qkv_n // 64, 8), dtype=torch.int64, device=dev),
                 (5, 1): torch.zeros((args.batch + 2 * cfg.inter // 64, 8), dtype=torch.int64, device=dev)}
    for _ in range(8):
        r.step()
    torch.cuda.synchronize()
    r.check()
    for (li, site), clk in sorted(r.ngf_clk.items()):
        c = clk.cpu().numpy().astype(np.float64) / 100.0
        t0 = c[:, 0].min()
        rows, tiles = c[:args.batch] - t0, c[args.batch:] - t0
        q = lambda v: "%6.2f /%6.2f /%6.2f" % (v.min(), np.median(v), v.max())  # noqa: E731
        print("in-step layer %d %s (grid %d)" % (li, "norm -> gate_up" if site else "norm -> qkv", c.shape[0]))
        for k, what in ((0, "rows  start"), (5, "rows  inputs loaded"), (1, "rows  body done"), (2, "rows  published")):
            print("   %-22s %s" % (what, q(rows[:, k])))
        for k, what in ((0, "tiles start"), (1, "tiles ring requested"), (2, "tiles gate passed"), (3, "tiles K loop done"), (4, "tiles end")):
            print("   %-22s %s" % (what, q(tiles[:, k])))
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--context", type=int, default=1024)
    ap.add_argument("--group-size", type=int, default=-1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--levels", default="3,4")
    ap.add_argument("--timeline", action="store_true")
    ap.add_argument("--isolated", action="store_true")
    ap.add_argument("--no-step", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.timeline:
        timeline(args, dev)
        step_timeline(args, dev)
    if args.isolated:
        isolated(args, dev)
    if not args.no_step:
        step_ab(args, dev)


if __name__ == "__main__":
    main()
