"""Decode attention (bs 16, T 1024) with L2-warm KV pages (the same layer every launch) vs cold (rotating over the 32 layers'
pools, 570 MB): an upper bound for what prefetching the KV pages of a layer into L2 could buy."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd.backend import fused_ext
from omniserve_amd.runtime import DecodeRunner, LlamaConfig
dev = torch.device("cuda:0")
r = DecodeRunner(LlamaConfig.llama3_8b(-1), 16, 1024, 64, dev, seed=0, use_graph=False, fused=2)
r.step(); torch.cuda.synchronize()
c = r.cfg
q = r.qkv_buf[:, : c.heads * 128].view(16, c.heads, 128)
k = r.qkv_buf[:, c.heads * 128:(c.heads + c.kv_heads) * 128].view(16, c.kv_heads, 128)
v = r.qkv_buf[:, (c.heads + c.kv_heads) * 128:].view(16, c.kv_heads, 128)
def call(li):
    fused_ext.decode_attention_quant_fuse_sum(r._q_attn, q, k, v, r.block_tables[li], r.lengths, r.tpb, r.max_context,
                                              c.rope_theta, r.act_sum2, r.act_scale2)
for name, layers in (("cold (32 layers rotated)", list(range(32))), ("warm (layer 0 only)", [0] * 32)):
    for li in layers: call(li)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for li in layers: call(li)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): g.replay()
    e.record(); torch.cuda.synchronize()
    print("%s: %.2f us per (attention + merge/quant) pair" % (name, s.elapsed_time(e) / 320 * 1e3))
