#!/usr/bin/env python
"""KV4 decode attention of one layer at the headline shape (B = 16, 32 q / 8 kv heads), pools rotated over 32 layers (cold), as a
function of the KV split count: us per layer of the reference entry point (partials + merge kernel; nsplit = 1: one DIRECT launch).

    python tools/attn_split_probe.py [--context 1024] [--batch 16]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omniserve_backend.fused_attention_pure_dense as fa  # noqa: E402
from omniserve_amd import _lib  # noqa: E402
from omniserve_amd.runtime import DecodeRunner, LlamaConfig  # noqa: E402


def graph_time(fn, nl, reps=8):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(nl):
            fn(i)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(nl):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * nl) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--context", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--warm", type=int, default=0, help="rotate over this many layers' pools only (1: K / V stay L2-resident)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = LlamaConfig.llama3_8b(-1)
    r = DecodeRunner(cfg, args.batch, args.context, 8, dev, seed=3, use_graph=False, fused=0)
    r.lengths.add_(1)
    B, hq, hk, d, nl = r.B, r.hl, r.kl, cfg.head_dim, len(r.layers)
    q = r.qkv_buf[:, : hq * d].view(B, hq, d); k = r.qkv_buf[:, hq * d:(hq + hk) * d].view(B, hk, d)
    v = r.qkv_buf[:, (hq + hk) * d:].view(B, hk, d)
    r.qkv_buf.normal_()
    for ns in ((0, 2, 4) if args.warm else (0, 1, 2, 3, 4, 6, 8)):
        _lib.lib().omni_kv4_decode_set_split_override(ns)

        def fn(i):
            fa.single_query_attention(q, k, v, r.block_tables[i % (args.warm or nl)], r.lengths, None, 65536, r.tpb, hk * d // 2, r.max_context, d,
                                      cfg.rope_theta, True, True, True)
        us = graph_time(fn, nl)
        print("nsplit %s: %7.2f us per layer (B = %d, T = %d)" % ("auto" if ns == 0 else ns, us, B, args.context))
    _lib.lib().omni_kv4_decode_set_split_override(0)


if __name__ == "__main__":
    main()
