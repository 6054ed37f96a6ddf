"""Golden vectors of one QServe decoder layer produced by the REFERENCE's own model code.

    python tests/golden/make_golden_layer.py            # needs /root/reference (build container); writes
                                                        # tests/golden/decoder_layer_w4a8kv4.npz (per channel) and
                                                        # tests/golden/decoder_layer_w4a8kv4_g128.npz (BASELINE configs[2])

`omniserve/modeling/models/llama_w4a8_unpad.py::LlamaDecoderLayer` (unmodified, imported from /root/reference) is
instantiated with per-channel (group_size -1) or per-group (128: two-level QoQ weights, norms and quantisers without the
activation sum, llama_w4a8_unpad.py:81,212-215,395-401) W4A8 weights produced by the reference's own packer (`from_linear`) and driven exactly as
the engine drives it -- one context-stage call over two 70-token prompts, then two generation-stage calls -- with its
`omniserve_backend.*` calls landing on the oracle-backed C-ABI of tests/refstack.py (CPU).  What is recorded: the layer's
packed weights, its inputs, its hidden-state outputs and the KV4 pages it leaves behind (per sequence, in logical page
order).  tests/test_reference_layer_golden_gpu.py replays the same inputs through omniserve_amd.runtime.DecodeRunner on
the MI355X (HIP kernels, the runner's own call sequence incl. the fused entry points) and compares; the CPU test
tests/test_reference_layer_golden_cpu.py re-generates the vectors in memory when /root/reference is present, so a stale
file cannot hide.  The arithmetic under the reference's layer here is oracle/: what these vectors pin is the WIRING --
which buffer feeds which call in which order, residual handling, in-place RoPE, cache append, lengths -- of the
reference's model code against the runner's.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import refstack  # noqa: E402

HIDDEN, INTER, HQ, HK, D, TPB = 256, 512, 2, 1, 128, 64
ROPE_BASE, EPS = 500000.0, 1e-5
B, L, DEC_STEPS, PAGES = 2, 70, 2, 2
OUT = os.path.join(HERE, "decoder_layer_w4a8kv4.npz")
# "h4": four q heads on one kv head, hidden 512 -- the smallest layer the runner's fusion level 3 accepts (its wide attention
# merge hands one wave 4 heads, and the row-kernel-free projections want hidden / 64 >= batch): the vectors the headline
# fusion level is replayed against (tests/test_reference_layer_golden_gpu.py)
VARIANTS = {"base": (256, 512, 2, 1), "h4": (512, 1024, 4, 1)}


def out_path(group_size, variant="base"):
    tag = ("" if group_size == -1 else "_g%d" % group_size) + ("" if variant == "base" else "_" + variant)
    return os.path.join(HERE, "decoder_layer_w4a8kv4%s.npz" % tag)


def _sp_attn_config():
    ns = types.SimpleNamespace()
    ns.sparse_kv_cache_enabled = lambda: False
    ns.get_dec_sub_chunk_per_block = lambda: 4
    ns.get_sparse_decode_mode = lambda: 0
    ns.get_dec_dynamic_sparse_token_budget = lambda: 4096
    ns.get_dec_selector_update_interval = lambda: 4
    return ns


def generate(group_size=-1, variant="base"):
    global HIDDEN, INTER, HQ, HK
    keep = (HIDDEN, INTER, HQ, HK)
    HIDDEN, INTER, HQ, HK = VARIANTS[variant]
    try:
        return _generate(group_size)
    finally:
        HIDDEN, INTER, HQ, HK = keep


def _generate(group_size):
    from omniserve_amd import ckpt
    from oracle import kv4
    with refstack.reference_over_mirror():
        from omniserve.modeling.layers.quantized_linear.w4a8_linear import W4A8OF16LinearDynamicInputScale as RefLinear
        from omniserve.modeling.models.llama_w4a8_unpad import LlamaDecoderLayer
        import omniserve_backend.fused_attention_fine_grained_dense as fgd
        g = torch.Generator().manual_seed(20260924)
        cfg = types.SimpleNamespace(hidden_size=HIDDEN, intermediate_size=INTER, num_attention_heads=HQ,
                                    num_key_value_heads=HK, rope_theta=ROPE_BASE, rope_scaling=None,
                                    max_position_embeddings=8192, rms_norm_eps=EPS, attention_bias=False)
        model_config = types.SimpleNamespace(sp_attn_config=_sp_attn_config(), kv_quant_granularity="fine_grained",
                                             multiblock_switch=2048, chunk_prefill_size=1 << 20)
        kvcfg = {"INT4_ENABLED": True, "ZEROS_ENABLED": True}
        layer = LlamaDecoderLayer(cfg, model_config, group_size, 0, kvcfg)
        weights = {}

        def fill(dst, n, k, name, scale):
            w = torch.randn((n, k), generator=g) * scale
            fake, s1, s2, z = ckpt.qoq_quantize_weight(w, group_size)
            lin = torch.nn.Linear(k, n, bias=False)
            lin.weight.data = fake.clone()
            src = RefLinear.from_linear(lin, 4, group_size, s1_scale=s1.float(), s2_scale=s2, zeros=z)     # the reference's packer
            for b in (("qweight", "s1_scales", "s1_szeros") if group_size == -1 else ("qweight", "s1_scales", "s2_scales", "s2_zeros")):
                getattr(dst, b).data = getattr(src, b).data.clone()
                weights["%s.%s" % (name, b)] = getattr(src, b).data.numpy().copy()

        fill(layer.self_attn.qkv_proj, (HQ + 2 * HK) * D, HIDDEN, "qkv", 0.06)
        fill(layer.self_attn.o_proj, HIDDEN, HQ * D, "o", 0.05)
        fill(layer.mlp.gate_up_proj, 2 * INTER, HIDDEN, "gate_up", 0.06)
        fill(layer.mlp.down_proj, HIDDEN, INTER, "down", 0.05)
        for name, norm in (("ln1", layer.input_layernorm), ("ln2", layer.post_attention_layernorm)):
            norm.weight.data = (1.0 + 0.1 * torch.randn((HIDDEN,), generator=g)).half()
            weights[name] = norm.weight.data.numpy().copy()
        # what init_sparse_kv_cache / init_ctx_sparse_attn (ctx_attn_init.py:28-81) leave on a dense model
        at = layer.self_attn
        at.retrieval_head_flags = torch.ones((HK,), dtype=torch.int32)
        at.head_rank_table = torch.arange(HK, dtype=torch.int32)
        at.pooling_heads_idx = torch.arange(HK, dtype=torch.int32)
        at.num_retrieval_kv_heads, at.num_streaming_kv_heads = HK, 0
        at.sink_blocks = at.local_blocks = at.sink_size = at.local_size = 0
        at.head_mask_type = at.streaming_info = None

        pb = kv4.page_bytes(HK, D, TPB)
        kpool = torch.zeros((B * PAGES, pb), dtype=torch.uint8)
        vpool = torch.zeros((B * PAGES, pb), dtype=torch.uint8)
        rng = np.random.default_rng(3)
        kid = rng.permutation(B * PAGES).reshape(B, PAGES)
        vid = rng.permutation(B * PAGES).reshape(B, PAGES)
        tab = torch.stack([kpool.data_ptr() + torch.from_numpy(kid) * pb, vpool.data_ptr() + torch.from_numpy(vid) * pb],
                          dim=1).to(torch.int64).contiguous()

        def buffers(T):
            f16, i8 = torch.float16, torch.int8
            return types.SimpleNamespace(
                batched_seq_len=T, hidden_size=HIDDEN, intermediate_size=INTER,
                quantized_hidden_states_buffer=torch.empty((T, HIDDEN), dtype=i8),
                quantized_scale_buffer=torch.empty((T,), dtype=f16), quantized_sum_buffer=torch.empty((T,), dtype=f16),
                qkv_proj_act_buffer=torch.empty((T, (HQ + 2 * HK) * D), dtype=f16),
                out_down_proj_act_buffer=torch.empty((T, HIDDEN), dtype=f16),
                gate_up_proj_act_buffer=torch.empty((T, 2 * INTER), dtype=f16),
                quantized_mlp_act_buffer=torch.empty((T, INTER), dtype=i8))

        def pages():
            k = np.stack([kpool.numpy()[kid[b]] for b in range(B)])     # [B, PAGES, page bytes], logical order
            v = np.stack([vpool.numpy()[vid[b]] for b in range(B)])
            return k.copy(), v.copy()

        out = dict(weights)
        # ---- context stage: two prompts of L tokens (model_runner.py builds cu_seqlens / padding offsets like this)
        T = B * L
        x = (torch.randn((T, HIDDEN), generator=g) * 0.8).half()
        cu = torch.arange(0, B + 1, dtype=torch.int32) * L
        lens = torch.full((B,), L, dtype=torch.int32)
        meta = types.SimpleNamespace(is_prompt=True, activation_buffer=buffers(T), cu_seqlens=cu, max_seq_len=L,
                                     retrieval_context_lens=lens, streaming_context_lens=lens,
                                     padding_offsets=fgd.compute_padding_offsets(cu, L, T),
                                     retrieval_block_tables=[tab], streaming_block_tables=[None])
        out["prefill_in"] = x.numpy().copy()
        y = layer(x, meta)
        out["prefill_out"] = y.numpy().copy()
        out["prefill_k_pages"], out["prefill_v_pages"] = pages()
        # ---- generation stage: DEC_STEPS tokens per sequence (lengths include the new token, decoding_attention.py:156)
        for s in range(DEC_STEPS):
            xd = (torch.randn((B, HIDDEN), generator=g) * 0.8).half()
            dl = torch.full((B,), L + s + 1, dtype=torch.int32)
            meta = types.SimpleNamespace(is_prompt=False, activation_buffer=buffers(B), max_seq_len=int(dl.max()),
                                         retrieval_context_lens=dl, streaming_context_lens=dl,
                                         retrieval_block_tables=[tab], streaming_block_tables=[None])
            out["decode%d_in" % s] = xd.numpy().copy()
            out["decode%d_out" % s] = layer(xd, meta).numpy().copy()
            out["decode%d_k_pages" % s], out["decode%d_v_pages" % s] = pages()
        out["shape"] = np.asarray([HIDDEN, INTER, HQ, HK, D, TPB, B, L, DEC_STEPS, PAGES], np.int64)
        out["rope_base_eps"] = np.asarray([ROPE_BASE, EPS], np.float64)
        out["group_size"] = np.asarray([group_size], np.int64)
        return out


if __name__ == "__main__":
    for var in VARIANTS:
        for gs in (-1, 128):
            vec = generate(gs, var)
            np.savez_compressed(out_path(gs, var), **vec)
            print("wrote", out_path(gs, var), os.path.getsize(out_path(gs, var)), "bytes;", ", ".join(sorted(vec)))
