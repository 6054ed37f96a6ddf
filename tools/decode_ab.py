"""A/B of the decode step (BASELINE configs[1] by default) over the L2-prefetch knobs, in ONE process:
    python tools/decode_ab.py [--batch 16] [--context 1024] [--group-size -1] [--steps 64] [--layers 32]
Prints ms/step for prefetch off and for each (budget MiB, weight policy, blocks) combination; every variant must
produce the same tokens as the baseline (the prefetch is a hint)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd.runtime import DecodeRunner, LlamaConfig  # noqa: E402


def run(cfg, args, dev, **kw):
    r = DecodeRunner(cfg, args.batch, args.context, 3 * args.steps + 16, dev, seed=1234, fused=args.fused, **kw)
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
    toks = r.tokens.clone()
    del r
    torch.cuda.empty_cache()
    return best * 1e3, toks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--context", type=int, default=1024)
    ap.add_argument("--group-size", type=int, default=-1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--fused", type=int, default=2)
    ap.add_argument("--model", default="llama3_8b")
    ap.add_argument("--budgets", default="8,16,24,32")
    ap.add_argument("--blocks", default="240")
    ap.add_argument("--policies", default="1,0")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = getattr(LlamaConfig, args.model)(args.group_size)
    cfg.layers = args.layers
    base, toks0 = run(cfg, args, dev, prefetch_mb=0)
    print("prefetch off                      : %.4f ms/step  %.1f tok/s" % (base, args.batch / base * 1e3), flush=True)
    for blocks in [int(x) for x in args.blocks.split(",")]:
        for pol in [int(x) for x in args.policies.split(",")]:
            for mb in [float(x) for x in args.budgets.split(",")]:
                ms, toks = run(cfg, args, dev, prefetch_mb=mb, prefetch_blocks=blocks, weight_policy=pol)
                same = bool(torch.equal(toks, toks0))
                print("budget %5.1f MiB policy %d blocks %3d : %.4f ms/step  %.1f tok/s  (x%.3f)  tokens %s" % (
                    mb, pol, blocks, ms, args.batch / ms * 1e3, base / ms, "same" if same else "DIFFER"), flush=True)


if __name__ == "__main__":
    main()
