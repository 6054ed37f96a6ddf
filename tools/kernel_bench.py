"""Per-kernel measurements for every row of SURVEY.md section 8(a) on one MI355X (synthetic inputs, event-timed on
torch's current stream, everything through the omniserve_backend mirrors = the C ABI).  One JSON line per case:
algorithmic bytes / ops, time, achieved GB/s or TOP/s and the fraction of the bounding peak.
Usage (GPU box): python tools/kernel_bench.py > gpurun_out/kernel_bench.jsonl"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402

if os.environ.get("OMNI_TUNE_LIB"):     # A/B against a library variant (tools/build_variant.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["OMNI_TUNE_LIB"])
import omniserve_backend.activation_ops as act  # noqa: E402
import omniserve_backend.fused_attention_ctx_pool as ctx_pool  # noqa: E402
import omniserve_backend.fused_attention_fine_grained_dense as fgd  # noqa: E402
import omniserve_backend.fused_attention_fine_grained_sparse as fgs  # noqa: E402
import omniserve_backend.fused_attention_per_tensor_dense as ptd  # noqa: E402
import omniserve_backend.fused_attention_per_tensor_sparse as pts  # noqa: E402
import omniserve_backend.fused_attention_pure_dense as pd  # noqa: E402
import omniserve_backend.fused_attention_selector as selector  # noqa: E402
import omniserve_backend.fused_kernels as fk  # noqa: E402
import omniserve_backend.layernorm_ops as ln  # noqa: E402
import omniserve_backend.qgemm_w4a8_per_chn as g_chn  # noqa: E402
import omniserve_backend.qgemm_w4a8_per_group as g_grp  # noqa: E402
import omniserve_backend.qgemm_w8a8 as g_w8  # noqa: E402
from block_sparse_attn import flash_attn_varlen_func, token_streaming_attn_func  # noqa: E402

dev = torch.device("cuda:0")
HBM, INT8, FP16 = 8000.0, 5000.0, 2500.0   # GB/s, TOP/s, TFLOP/s (dense peaks, MI355X_MICROARCH.md)
D = 128


def timed(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def report(row, name, us, nbytes=None, ops=None, peak_kind="hbm"):
    r = {"row": row, "case": name, "us": round(us, 2)}
    if nbytes is not None:
        r["alg_bytes"] = int(nbytes)
        r["GBps"] = round(nbytes / us / 1e3, 1)
    if ops is not None:
        r["ops"] = int(ops)
        r["Tops"] = round(ops / us / 1e6, 1)
    if peak_kind == "hbm":
        r["frac_of_hbm_peak"] = round(r["GBps"] / HBM, 3)
    elif peak_kind == "int8":
        r["frac_of_int8_mfma_peak"] = round(r["Tops"] / INT8, 3)
    elif peak_kind == "fp16":
        r["frac_of_fp16_mfma_peak"] = round(r["Tops"] / FP16, 3)
    print(json.dumps(r), flush=True)


def rand_i8(*shape):
    return torch.randint(-127, 128, shape, dtype=torch.int8, device=dev)


def rand_w4(n, k):
    return torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, device=dev).view(torch.int8)


def gemms():
    for (M, N, K) in [(16, 28672, 4096), (64, 28672, 4096), (4096, 4096, 4096), (16384, 28672, 4096)]:
        a, w = rand_i8(M, K), rand_w4(N, K)
        sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = sw.clone()
        sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = sa.clone()
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        nb = M * K + N * K // 2 + 2 * M * N + 4 * N + 4 * M
        us = timed(lambda: g_chn.gemm_forward_cuda(a, w, sw, sa, sz, asum, out))
        report("a1", "w4a8 per-channel M=%d N=%d K=%d (L2/MALL-warm)" % (M, N, K), us, nb, 2.0 * M * N * K,
               "hbm" if M <= 128 else "int8")
        s2s = torch.randint(1, 9, (K // 128, N), dtype=torch.int8, device=dev)
        s2z = torch.randint(-100, 1, (K // 128, N), dtype=torch.int8, device=dev)
        us = timed(lambda: g_grp.gemm_forward_cuda(a, w, s2z, s2s, sw, sa, out))
        report("a2", "w4a8 g128 M=%d N=%d K=%d" % (M, N, K), us, nb + 2 * N * K // 128, 2.0 * M * N * K,
               "hbm" if M <= 128 else "int8")
        w8 = rand_i8(N, K)
        us = timed(lambda: g_w8.w8a8_gemm_forward_cuda(a, w8, sw, sa, out))
        report("a3", "w8a8 M=%d N=%d K=%d" % (M, N, K), us, M * K + N * K + 2 * M * N + 2 * N + 2 * M, 2.0 * M * N * K,
               "hbm" if M <= 128 else "int8")
        del a, w, w8, out


def row_kernels():
    for M in (16, 16384):
        H, I = 4096, 14336
        x = torch.randn((M, H), dtype=torch.float16, device=dev)
        g = torch.ones((H,), dtype=torch.float16, device=dev)
        q = torch.empty((M, H), dtype=torch.int8, device=dev)
        s = torch.empty((M,), dtype=torch.float16, device=dev); sm = s.clone()
        us = timed(lambda: ln.rms_norm_general_fuse_sum(q, x, g, sm, s, 1e-5, True))
        report("a4", "rms_norm_general_fuse_sum M=%d H=%d" % (M, H), us, M * H * 3 + 2 * H + 4 * M)
        us = timed(lambda: fk.invoke_quant_fuse_sum(q, x, sm, s))
        report("a5", "invoke_quant_fuse_sum M=%d H=%d" % (M, H), us, M * H * 3 + 4 * M)
        gu = torch.randn((M, 2 * I), dtype=torch.float16, device=dev)
        o = torch.empty((M, I), dtype=torch.float16, device=dev)
        us = timed(lambda: act.silu_and_mul(o, gu))
        report("a6", "silu_and_mul M=%d I=%d" % (M, I), us, M * I * 6)
        del x, gu, o


class Pools:
    """Synthetic KV4 pools + pointer tables ([B,2,blocks]) for `heads` heads per page."""

    def __init__(self, B, blocks, heads, tpb=64, stats_sub=0, row=64):
        """row = bytes of one token row of one head: 64 (KV4) or 128 (per-tensor KV8)."""
        self.page_bytes = heads * tpb * row + 2 * heads * tpb * 2
        kbytes = self.page_bytes + (2 * (tpb // stats_sub) * heads * D * 2 if stats_sub else 0)
        n = B * blocks
        self.k = torch.randint(0, 256, (n, kbytes), dtype=torch.uint8, device=dev)
        self.v = torch.randint(0, 256, (n, self.page_bytes), dtype=torch.uint8, device=dev)
        for pool in (self.k, self.v):     # sane fp16 scales / zeros
            tail = pool[:, heads * tpb * row: self.page_bytes].view(torch.float16).view(n, 2, heads * tpb)
            tail[:, 0] = 0.05
            tail[:, 1] = 7.5
        if stats_sub:
            self.k[:, self.page_bytes:].view(torch.float16).fill_(0.5)
        perm = torch.randperm(n, device=dev).view(B, blocks)
        self.table = torch.empty((B, 2, blocks), dtype=torch.int64, device=dev)
        self.table[:, 0] = self.k.data_ptr() + perm * kbytes
        self.table[:, 1] = self.v.data_ptr() + perm * self.page_bytes


def kv_kernels():
    Hq, Hk = 32, 8
    # a7: prefill writer, 16 x 1024 tokens
    B, L = 16, 1024
    T = B * L
    pools = Pools(B, L // 64 + 1, Hk)
    qkv = torch.randn((T, (Hq + 2 * Hk) * D), dtype=torch.float16, device=dev)
    lens = torch.full((B,), L, dtype=torch.int32, device=dev)
    cu = torch.arange(0, B + 1, dtype=torch.int32, device=dev) * L
    pad = pd.compute_padding_offsets(cu, L, T)
    flags = torch.ones((Hk,), dtype=torch.int32, device=dev); rank = torch.arange(Hk, dtype=torch.int32, device=dev)
    us = timed(lambda: fgd.apply_bias_rope_update_kv_cache(
        qkv, lens, None, pad, pools.table, None, flags, rank, Hq, Hk, L, 64, Hk * D // 2, 0, 0, 0, 0, 0, Hk, 0, D,
        500000.0, 1.0, 1 << 20, True, True, True))
    report("a7", "KV4 prefill writer %d tokens (RoPE q,k in place + quantise k,v)" % T, us,
           T * ((Hq + Hk) * 256 * 2 + Hk * 256 + 2 * Hk * 68))
    us = timed(lambda: pd.compute_padding_offsets(cu, L, T))
    report("a14", "compute_padding_offsets %d tokens" % T, us, T * 4)
    # a9: dense decode attention
    for (B, Tc) in [(16, 1024), (64, 1024), (8, 32768)]:
        pools = Pools(B, Tc // 64 + 2, Hk)
        lens = torch.full((B,), Tc + 1, dtype=torch.int32, device=dev)
        q = torch.randn((B, Hq, D), dtype=torch.float16, device=dev)
        k = torch.randn((B, Hk, D), dtype=torch.float16, device=dev); v = torch.randn_like(k)
        us = timed(lambda: pd.single_query_attention(q, k, v, pools.table, lens, None, 1 << 20, 64, Hk * D // 2, Tc + 1,
                                                     D, 500000.0, True, True, True))
        report("a9", "KV4 decode attention B=%d T=%d (GQA 32/8), incl. merge kernel" % (B, Tc), us, 1088 * Tc * B)
    # a10-a12: LServe config-4-like: B=1, T=256K, 4 retrieval + 4 streaming kv heads, 64-page budget, sub-chunk 16
    B, Tc, sub, budget = 1, 256000, 16, 64
    nr = ns = 4
    flags = torch.tensor([1, 0, 1, 0, 1, 0, 1, 0], dtype=torch.int32, device=dev)
    rank = torch.tensor([0, 0, 1, 1, 2, 2, 3, 3], dtype=torch.int32, device=dev)
    blocks = Tc // 64 + 2
    retr = Pools(B, blocks, nr, stats_sub=sub)
    strm = Pools(B, 2 + 5, ns)
    lens = torch.full((B,), Tc + 1, dtype=torch.int32, device=dev)
    q = torch.randn((B, Hq, D), dtype=torch.float16, device=dev)
    k = torch.randn((B, Hk, D), dtype=torch.float16, device=dev); v = torch.randn_like(k)
    common = (64, nr * D // 2, ns * D // 2, 128, 256, 2, 5, nr, ns, Tc + 1, D, 500000.0, 1.0, True, True, True)
    us = timed(lambda: fgd.single_query_attention(q, k, v, retr.table, strm.table, flags, rank, lens, None, 1 << 20,
                                                  *common, 2048), iters=5)
    report("a10", "fine-grained dense decode B=1 T=256000 (4 retrieval + 4 streaming kv heads)", us,
           544 * nr * Tc / 4 * 1 + 544 * ns * 383 / 4)
    dyn = torch.randint(0, Tc // 64 - 1, (B, Hq, budget), dtype=torch.int32, device=dev)
    dyn[..., -1] = (Tc - 1) // 64
    us = timed(lambda: fgs.single_query_attention(q, k, v, retr.table, strm.table, flags, rank, dyn, lens, None, 1 << 20,
                                                  *common, sub, nr * D, 2048))
    report("a10", "fine-grained sparse decode B=1 T=256000, 64-page budget per q head", us,
           16 * budget * 64 * 136 + 544 * ns * 383 / 4)
    us = timed(lambda: selector.single_query_page_selector(
        q, k, v, retr.table, strm.table, flags, rank, None, lens, None, 1 << 20, 64, nr * D // 2, ns * D // 2, 128, 256,
        2, 5, nr, ns, Tc, D, 500000.0, 1.0, True, True, True, sub, nr * D, 2048))
    report("a12", "page selector B=1 T=256000 sub-chunk 16 (16 retrieval q heads)", us, (Tc // sub) * nr * 2 * D * 2)
    Lp = 65536
    kk = torch.randn((Lp, Hk, D), dtype=torch.float16, device=dev)
    cu = torch.tensor([0, Lp], dtype=torch.int32, device=dev)
    heads_idx = torch.tensor([0, 2, 4, 6], dtype=torch.int32, device=dev)
    us = timed(lambda: ctx_pool.paged_min_max_pool(kk, retr.table, cu, heads_idx, Lp, sub, 64, nr * D // 2, True))
    report("a11", "paged_min_max_pool L=%d, 4 pooled heads, sub-chunk 16" % Lp, us, Lp * nr * D * 2 + (Lp // sub) * nr * D * 4)


def kv8_kernels():
    """the per-tensor KV8 family (LServe's published w8a8kv8 configuration)."""
    Hq, Hk = 32, 8
    qo = torch.tensor([0.03, 0.035], dtype=torch.float32, device=dev)
    oq = 1.0 / qo
    B, L = 16, 1024
    T = B * L
    pools = Pools(B, L // 64 + 1, Hk, row=128)
    qkv = torch.randn((T, (Hq + 2 * Hk) * D), dtype=torch.float16, device=dev)
    lens = torch.full((B,), L, dtype=torch.int32, device=dev)
    cu = torch.arange(0, B + 1, dtype=torch.int32, device=dev) * L
    pad = pd.compute_padding_offsets(cu, L, T)
    flags = torch.ones((Hk,), dtype=torch.int32, device=dev); rank = torch.arange(Hk, dtype=torch.int32, device=dev)
    us = timed(lambda: ptd.apply_bias_rope_update_kv_cache(
        qkv, oq, lens, None, pad, pools.table, None, flags, rank, Hq, Hk, L, 64, Hk * D, 0, 0, 0, 0, 0, Hk, 0, D,
        500000.0, 1.0, 1 << 20, True, False, False))
    report("f2", "KV8 per-tensor prefill writer %d tokens" % T, us, T * ((Hq + Hk) * 256 * 2 + Hk * 256 + 2 * Hk * 130))
    for (B, Tc) in [(16, 1024), (8, 32768)]:
        pools = Pools(B, Tc // 64 + 2, Hk, row=128)
        lens = torch.full((B,), Tc + 1, dtype=torch.int32, device=dev)
        q = torch.randn((B, Hq, D), dtype=torch.float16, device=dev)
        k = torch.randn((B, Hk, D), dtype=torch.float16, device=dev); v = torch.randn_like(k)
        us = timed(lambda: ptd.single_query_attention(q, k, v, qo, oq, pools.table, None, flags, rank, lens, None,
                                                      1 << 20, 64, Hk * D, 0, 0, 0, 0, 0, Hk, 0, Tc + 1, D, 500000.0,
                                                      1.0, True, False, False, 2048))
        report("f2", "KV8 per-tensor decode attention B=%d T=%d (GQA 32/8, all retrieval heads)" % (B, Tc), us,
               2 * Hk * D * Tc * B)
    B, Tc, sub, budget = 1, 256000, 16, 64
    nr = ns = 4
    flags = torch.tensor([1, 0, 1, 0, 1, 0, 1, 0], dtype=torch.int32, device=dev)
    rank = torch.tensor([0, 0, 1, 1, 2, 2, 3, 3], dtype=torch.int32, device=dev)
    retr = Pools(B, Tc // 64 + 2, nr, stats_sub=sub, row=128)
    strm = Pools(B, 2 + 5, ns, row=128)
    lens = torch.full((B,), Tc + 1, dtype=torch.int32, device=dev)
    q = torch.randn((B, Hq, D), dtype=torch.float16, device=dev)
    k = torch.randn((B, Hk, D), dtype=torch.float16, device=dev); v = torch.randn_like(k)
    common = (64, nr * D, ns * D, 128, 256, 2, 5, nr, ns, Tc + 1, D, 500000.0, 1.0, True, False, False)
    us = timed(lambda: ptd.single_query_attention(q, k, v, qo, oq, retr.table, strm.table, flags, rank, lens, None,
                                                  1 << 20, *common, 2048), iters=5)
    report("f2", "KV8 per-tensor dense decode B=1 T=256000 (4 retrieval + 4 streaming kv heads)", us,
           2 * D * nr * Tc + 2 * D * ns * 383)
    dyn = torch.randint(0, Tc // 64 - 1, (B, Hq, budget), dtype=torch.int32, device=dev)
    dyn[..., -1] = (Tc - 1) // 64
    us = timed(lambda: pts.single_query_attention(q, k, v, qo, oq, retr.table, strm.table, flags, rank, dyn, lens, None,
                                                  1 << 20, *common, sub, nr * D, 2048))
    report("f2", "KV8 per-tensor sparse decode B=1 T=256000, 64-page budget per q head", us,
           16 * budget * 64 * 256 + 2 * D * ns * 383)
    us = timed(lambda: selector.single_query_page_selector(
        q, k, v, retr.table, strm.table, flags, rank, None, lens, None, 1 << 20, 64, nr * D, ns * D, 128, 256,
        2, 5, nr, ns, Tc, D, 500000.0, 1.0, True, False, True, sub, nr * D, 2048))
    report("f2", "page selector on KV8 pages B=1 T=256000 sub-chunk 16", us, (Tc // sub) * nr * 2 * D * 2)


def prefill_attention():
    Hq, Hk = 32, 8
    for L in (4096, 16384):
        qkv = torch.randn((L, (Hq + 2 * Hk) * D), dtype=torch.float16, device=dev)
        q = qkv[:, : Hq * D].view(L, Hq, D)
        k = qkv[:, Hq * D:(Hq + Hk) * D].view(L, Hk, D)
        v = qkv[:, (Hq + Hk) * D:].view(L, Hk, D)
        cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
        us = timed(lambda: flash_attn_varlen_func(q, k, v, cu, cu, L, L, causal=True), iters=3, warm=1)
        report("a13", "prefill attention dense causal L=%d (32 q / 8 kv heads)" % L, us, None, 4.0 * L * L / 2 * D * Hq, "fp16")
        hm = torch.tensor([0, -1] * (Hq // 2), dtype=torch.int32, device=dev)
        si = torch.tensor([128, 8192] * Hq, dtype=torch.int32, device=dev)
        us = timed(lambda: token_streaming_attn_func(q, k, v, cu, cu, hm, si, L, L), iters=3, warm=1)
        win = min(L, 128 + 8192)
        report("a13", "prefill attention 16 dense + 16 streaming (sink 128, local 8192) heads L=%d" % L, us, None,
               4.0 * D * (Hq // 2) * (L * L / 2 + L * win - (win * win / 2 if L > win else L * L / 2)), "fp16")


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "row", "kv", "kv8", "attn"]
    if "gemm" in which:
        gemms()
    if "row" in which:
        row_kernels()
    if "kv" in which:
        kv_kernels()
    if "kv8" in which:
        kv8_kernels()
    if "attn" in which:
        prefill_attention()
