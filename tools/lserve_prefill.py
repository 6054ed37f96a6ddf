"""Context stage (TTFT) of the LServe configuration through LServeDecodeRunner.prefill: Llama-3-8B W8A8, batch 1, 4 + 4 kv
heads (retrieval + streaming), sink 128 / local 8192 in the context stage, per-tensor KV8 or fine-grained KV4 pages.
    python tools/lserve_prefill.py [kv8|kv4] LEN [LEN ...]
Prints seconds per prompt, tokens/s and the int8-GEMM / attention operation counts behind them."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd.lserve_runtime import LServeDecodeRunner  # noqa: E402
from omniserve_amd.runtime import LlamaConfig  # noqa: E402

fmt = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("kv8", "kv4") else "kv8"
lens = [int(a) for a in sys.argv[1:] if a.isdigit()] or [16384]
dev = torch.device("cuda:0")
cfg = LlamaConfig.llama3_8b(-1)
r = LServeDecodeRunner(cfg, 1, max(lens), 16, dev, seed=7, kv_format=fmt)
r.prefill(seq_len=2048)          # sizes scratch, builds RoPE tables
torch.cuda.synchronize()
for L in lens:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = r.prefill(seq_len=L)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(x[-1].float()).all()
    gemm = 2.0 * L * sum(Ly[k].weight.numel() for Ly in r.layers for k in ("qkv", "o", "gate_up", "down"))
    win = 128 + 8192
    dense = L * L / 2.0
    strm = dense if L <= win else L * win - win * win / 2.0
    attn = 4.0 * cfg.head_dim * (cfg.heads // 2) * (dense + strm) * cfg.layers
    print("%s L=%7d: %7.3f s  %9.0f tok/s   int8 GEMM %.2f POP (%.0f TOPS if alone)  attention %.2f PFLOP (%.0f TF if alone)"
          % (fmt, L, dt, L / dt, gemm * 1e-15, gemm / dt * 1e-12, attn * 1e-15, attn / dt * 1e-12), flush=True)
    for _ in range(4):           # and the sequence decodes on from there
        r.step()
    torch.cuda.synchronize()
    assert torch.isfinite(r.x.float()).all() and int(r.lengths[0]) == L + 4
