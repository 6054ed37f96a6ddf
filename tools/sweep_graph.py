"""Graph-timed tuning sweeps on the GPU (no host launch overhead in the numbers):
  * decode W4A8 GEMM plans (waves per workgroup, K splits) on the Llama-3-8B projection shapes,
  * KV split count of the decode attention at bs=16 / context 1024.
Usage: python tools/sweep_graph.py [M] > gpurun_out/sweep_graph.log"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402
from omniserve_amd.backend import qgemm_w4a8_per_chn  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib()


def graph_time_us(fn, n_inner, reps=20):
    """fn(i) enqueues launch i; capture n_inner launches in a graph, replay `reps` times."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(n_inner):
            fn(i)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n_inner):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * n_inner)


def sweep_gemm(M):
    shapes = [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]
    for (N, K) in shapes:
        copies = max(4, int(700e6 // (N * K // 2)))
        ws = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8) for _ in range(copies)]
        a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
        sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = sw.clone()
        sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = sa.clone()
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        alg = M * K + N * K // 2 + 2 * M * N + 4 * N + 4 * M
        best = None
        for waves in (4, 2, 1):
            for sk in (1, 2, 4, 7, 8, 14, 16, 28, 32):
                if K % (sk * 64) or K // sk < 128 or (16 * ((M + 15) // 16)) * (K // sk) > 65536:
                    continue
                lib.omni_gemm_set_plan_override(waves, sk)
                us = graph_time_us(lambda i: qgemm_w4a8_per_chn.gemm_forward_cuda(a, ws[i % copies], sw, sa, sz, asum, out), copies)
                print("gemm M=%d N=%d K=%d waves=%d sk=%2d : %7.2f us  %7.1f GB/s" % (M, N, K, waves, sk, us, alg / us / 1e3), flush=True)
                if best is None or us < best[0]:
                    best = (us, waves, sk)
        lib.omni_gemm_set_plan_override(0, 0)
        us = graph_time_us(lambda i: qgemm_w4a8_per_chn.gemm_forward_cuda(a, ws[i % copies], sw, sa, sz, asum, out), copies)
        print("gemm M=%d N=%d K=%d heuristic : %7.2f us | best %s" % (M, N, K, us, best), flush=True)
        del ws


def sweep_attn(batch=16, context=1024):
    from omniserve_amd.runtime import DecodeRunner, LlamaConfig
    from omniserve_amd.backend import fused_attention_pure_dense as fa
    cfg = LlamaConfig.llama3_8b()
    cfg.layers = 12
    cfg.vocab = 1024
    r = DecodeRunner(cfg, batch, context, 8, dev, seed=0, use_graph=False)
    B, hq, hk, d = batch, cfg.heads, cfg.kv_heads, cfg.head_dim
    r.qkv_buf.normal_()
    q = r.qkv_buf[:, : hq * d].view(B, hq, d)
    k = r.qkv_buf[:, hq * d:(hq + hk) * d].view(B, hk, d)
    v = r.qkv_buf[:, (hq + hk) * d:].view(B, hk, d)
    lens = torch.full((B,), context + 1, dtype=torch.int32, device=dev)
    kvb = 2 * (hk * d // 2 + hk * 4) * context * B
    for ns in (0, 1, 2, 3, 4, 6, 8):
        lib.omni_kv4_decode_set_split_override(ns)
        us = graph_time_us(lambda i: fa.single_query_attention(q, k, v, r.block_tables[i % cfg.layers], lens, None, 65536, 64,
                                                               hk * d // 2, context + 8, d, cfg.rope_theta, True, True, True),
                           cfg.layers)
        print("attn B=%d ctx=%d nsplit=%d : %7.2f us  %7.1f GB/s" % (B, context, ns, us, kvb / us / 1e3), flush=True)
    lib.omni_kv4_decode_set_split_override(0)


if __name__ == "__main__":
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    sweep_attn()
    sweep_gemm(M)
