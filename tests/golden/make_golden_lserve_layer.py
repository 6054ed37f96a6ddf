"""Golden vectors of one LServe decoder layer produced by the REFERENCE's own model code.

    python tests/golden/make_golden_lserve_layer.py     # needs /root/reference (build container); writes
                                                        # tests/golden/lserve_layer_{kv8,kv4}.npz

`omniserve/modeling/models/llama_w8a8_unpad.py::LlamaDecoderLayer` (unmodified, imported from /root/reference) is built
with the LServe configuration of BASELINE configs[3] in miniature -- W8A8 linears, half of the kv heads streaming
(`attn_config.sparse_attn_init` on a two-head pattern file, `ctx_attn_init.init_sparse_kv_cache / init_ctx_sparse_attn`,
all the reference's own), sparse context attention, dynamic sparse decoding (page selector every 2nd step, 2 + 1 pages of
budget, 4 sub-chunks per page) -- once with per_tensor KV8 pages (scripts/lserve_benchmark/launch.sh) and once with
fine_grained KV4 pages, and driven as the engine drives it: one context-stage call over two 318-token prompts, then four
generation-stage calls (the third crosses a page boundary; selection refreshes on steps 1, 2, 4).  Its `omniserve_backend.*`
/ `block_sparse_attn` calls land on the oracle-backed C-ABI of tests/refstack.py (CPU).  Recorded: weights, inputs, hidden
outputs, the selected pages after every step and both page pools (retrieval K with statistics / V, streaming ring K / V, in
table order) after the context stage and after every step.  tests/test_reference_lserve_layer_golden_gpu.py replays them
through omniserve_amd.lserve_runtime.LServeDecodeRunner on the MI355X; tests/test_reference_layer_golden_cpu.py
re-generates them in memory when /root/reference is present.  As in make_golden_layer.py the arithmetic under the
reference's layer is oracle/: what the vectors pin is the WIRING of the runner against the reference's model code --
operands, order, lengths / timestep conventions, when the page selection refreshes, which pages each head class reads.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import refstack  # noqa: E402

HIDDEN, INTER, HQ, HK, D, TPB = 512, 1024, 4, 2, 128, 64
ROPE_BASE, EPS = 500000.0, 1e-5
B, L, DEC_STEPS = 2, 318, 4
CTX_SINK, CTX_LOCAL, DEC_SINK, DEC_LOCAL = 64, 128, 64, 128
SUBS, BUDGET, INTERVAL = 4, 128, 2
KV_SCALES = (0.03, 0.035)
RPAGES = (L + DEC_STEPS) // TPB + 1


def out_path(fmt):
    return os.path.join(HERE, "lserve_layer_%s.npz" % fmt)


def generate(fmt):
    assert fmt in ("kv8", "kv4")
    from oracle import kv4, kv8
    with refstack.reference_over_mirror(), tempfile.TemporaryDirectory() as pattern_dir:
        from omniserve.attn_config import sparse_attn_init
        from omniserve.modeling.layers.ctx_attn.ctx_attn_init import init_ctx_sparse_attn, init_sparse_kv_cache
        from omniserve.modeling.models.llama_w8a8_unpad import LlamaDecoderLayer
        import omniserve_backend.fused_attention_fine_grained_dense as fgd
        g = torch.Generator().manual_seed(20260925)
        # layer 0: head 0 = retrieval (dense) head, head 1 = streaming head -- a DuoAttention pattern file in miniature
        # (two layers: numpy reads a one-row file back as a vector)
        np.savetxt(os.path.join(pattern_dir, "full_attention_heads.tsv"), np.asarray([[0.9, 0.1], [0.2, 0.8]]), delimiter="\t")
        json.dump({}, open(os.path.join(pattern_dir, "config.json"), "w"))
        sp = sparse_attn_init(HK, 2, TPB, True, 1, pattern_dir, 0.5, CTX_SINK, CTX_LOCAL, DEC_SINK, DEC_LOCAL, SUBS, BUDGET,
                              INTERVAL)
        assert sp.get_full_attention_heads().tolist() == [[1, 0], [0, 1]]
        cfg = types.SimpleNamespace(hidden_size=HIDDEN, intermediate_size=INTER, num_attention_heads=HQ,
                                    num_key_value_heads=HK, rope_theta=ROPE_BASE, rope_scaling=None,
                                    max_position_embeddings=8192, rms_norm_eps=EPS, attention_bias=False)
        model_config = types.SimpleNamespace(sp_attn_config=sp, multiblock_switch=2048, chunk_prefill_size=1 << 20,
                                             kv_quant_granularity="per_tensor" if fmt == "kv8" else "fine_grained")
        kvcfg = {"INT4_ENABLED": fmt == "kv4", "ZEROS_ENABLED": fmt == "kv4"}
        layer = LlamaDecoderLayer(cfg, model_config, 0, kvcfg)
        out = {}
        for name, lin in (("qkv", layer.self_attn.qkv_proj), ("o", layer.self_attn.o_proj),
                          ("gate_up", layer.mlp.gate_up_proj), ("down", layer.mlp.down_proj)):
            n, k = lin.weight.shape
            lin.weight.data = torch.randint(-127, 128, (n, k), generator=g, dtype=torch.int8)
            lin.dequant_scale.data = ((torch.rand((n,), generator=g) * 0.9 + 0.35) / (127.0 * k ** 0.5)).to(lin.dequant_scale.dtype)
            out[name + ".weight"] = lin.weight.data.numpy().copy()
            out[name + ".dequant_scale"] = lin.dequant_scale.data.half().numpy().copy()
        for name, norm in (("ln1", layer.input_layernorm), ("ln2", layer.post_attention_layernorm)):
            norm.weight.data = (1.0 + 0.1 * torch.randn((HIDDEN,), generator=g)).half()
            out[name] = norm.weight.data.numpy().copy()
        layer.self_attn.kv_scale_quant_orig.data = torch.tensor(KV_SCALES)
        # the reference's own initialisers on a one-layer stand-in for LlamaForCausalLM
        model = types.SimpleNamespace(model=types.SimpleNamespace(layers=[layer]), total_num_heads=HQ, total_num_kv_heads=HK,
                                      parameters=lambda: iter([layer.input_layernorm.weight]))
        init_sparse_kv_cache(model, sp)
        init_ctx_sparse_attn(model, sp)
        at = layer.self_attn
        nr, ns = at.num_retrieval_kv_heads, at.num_streaming_kv_heads
        spages = at.sink_blocks + at.local_blocks
        out["head_setup"] = np.asarray([nr, ns, at.sink_size, at.local_size, at.sink_blocks, at.local_blocks], np.int64)
        out["retrieval_head_flags"] = at.retrieval_head_flags.numpy().astype(np.int32)
        out["head_mask_type"] = at.head_mask_type.numpy().astype(np.int32)
        out["streaming_info"] = at.streaming_info.numpy().astype(np.int32)

        # page pools as the cache engine allocates them (cache_engine.py:73-136): K pages of the retrieval pool carry statistics
        row = D if fmt == "kv8" else D // 2
        plain = lambda heads: heads * TPB * (row + 4)
        rk_bytes = plain(nr) + 2 * SUBS * nr * D * 2
        pools = dict(rk=torch.zeros((B * RPAGES, rk_bytes), dtype=torch.uint8), rv=torch.zeros((B * RPAGES, plain(nr)), dtype=torch.uint8),
                     sk=torch.zeros((B * spages, plain(ns)), dtype=torch.uint8), sv=torch.zeros((B * spages, plain(ns)), dtype=torch.uint8))
        rng = np.random.default_rng(5)
        ids = dict(rk=rng.permutation(B * RPAGES).reshape(B, RPAGES), rv=rng.permutation(B * RPAGES).reshape(B, RPAGES),
                   sk=rng.permutation(B * spages).reshape(B, spages), sv=rng.permutation(B * spages).reshape(B, spages))

        def table(k, v):
            return torch.stack([pools[k].data_ptr() + torch.from_numpy(ids[k]) * pools[k].shape[1],
                                pools[v].data_ptr() + torch.from_numpy(ids[v]) * pools[v].shape[1]], dim=1).to(torch.int64).contiguous()
        rtab, stab = table("rk", "rv"), table("sk", "sv")

        def snapshot(tag):
            for name in pools:
                out["%s_%s" % (tag, name)] = np.stack([pools[name].numpy()[ids[name][b]] for b in range(B)]).copy()

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

        # ---- context stage (model_runner.py:262-360)
        T = B * L
        x = (torch.randn((T, HIDDEN), generator=g) * 0.8).half()
        cu = torch.arange(0, B + 1, dtype=torch.int32) * L
        lens = torch.full((B,), L, dtype=torch.int32)
        slens = torch.full((B,), min(L, DEC_SINK + DEC_LOCAL), dtype=torch.int32)
        meta = types.SimpleNamespace(is_prompt=True, activation_buffer=buffers(T), cu_seqlens=cu, max_seq_len=L,
                                     retrieval_context_lens=lens, streaming_context_lens=slens,
                                     padding_offsets=fgd.compute_padding_offsets(cu, L, T),
                                     retrieval_block_tables=[rtab], streaming_block_tables=[stab])
        out["prefill_in"] = x.numpy().copy()
        out["prefill_out"] = layer(x, meta).numpy().copy()
        snapshot("prefill")
        # ---- generation stage (model_runner.py:368-445): lengths and max_seq_len count the token being generated
        for s in range(DEC_STEPS):
            xd = (torch.randn((B, HIDDEN), generator=g) * 0.8).half()
            n = L + s + 1
            dl = torch.full((B,), n, dtype=torch.int32)
            meta = types.SimpleNamespace(is_prompt=False, activation_buffer=buffers(B), max_seq_len=n,
                                         retrieval_context_lens=dl,
                                         streaming_context_lens=torch.full((B,), min(n, DEC_SINK + DEC_LOCAL), dtype=torch.int32),
                                         retrieval_block_tables=[rtab], streaming_block_tables=[stab])
            out["decode%d_in" % s] = xd.numpy().copy()
            out["decode%d_out" % s] = layer(xd, meta).numpy().copy()
            out["decode%d_pages" % s] = at.cached_dynamic_sparse_page_idx.numpy().astype(np.int32).copy()
            snapshot("decode%d" % s)
        out["shape"] = np.asarray([HIDDEN, INTER, HQ, HK, D, TPB, B, L, DEC_STEPS, RPAGES, spages, SUBS, BUDGET, INTERVAL,
                                   CTX_SINK, CTX_LOCAL], np.int64)
        out["floats"] = np.asarray([ROPE_BASE, EPS, KV_SCALES[0], KV_SCALES[1]], np.float64)
        return out


if __name__ == "__main__":
    for fmt in ("kv8", "kv4"):
        vec = generate(fmt)
        np.savez_compressed(out_path(fmt), **vec)
        print("wrote", out_path(fmt), os.path.getsize(out_path(fmt)), "bytes")
        print("  pages per step:", [vec["decode%d_pages" % s][0, 0].tolist() for s in range(DEC_STEPS)])
