"""Golden fixtures of the reference's HOST-side logic on the hot path, produced by running the reference's own Python.

Run in the build container only (needs /root/reference; it does not exist on the GPU box):

    python tests/golden/make_golden_host.py            # rewrites tests/golden/host_logic.json + quantizer.npz

The reference's `omniserve` package imports on this CPU-only box once `import omniserve_backend...` resolves to THIS
repo's mirror package (sys.path order below) and `torch.cuda.current_device` is stubbed (it is evaluated in a default
argument at class-creation time, w4a8_linear.py:24).  No kernel is launched: everything recorded here is torch / Python
logic that feeds the kernels --

  page_choice      DecodingAttentionWrapper.dynamic_select_topk_pages (decoding_attention.py:88-142), with the
                   selector kernel stubbed to return prepared scores: view / max over sub-chunks / topk / cat newest page
  ring_map         BaseBlockSpaceManager.allocate + append_slot with a streaming window (block_manager.py:140-218):
                   logical block -> physical block of the sink + local ring
  head_masks       init_ctx_sparse_attn / init_sparse_kv_cache (ctx_attn_init.py:11-83): head_mask_type, streaming_info,
                   retrieval_head_flags, head_rank_table, pooling_heads_idx
  page_bytes       BaseCacheEngine.__init__ (cache_engine.py:41-88): bytes of a K / V page with and without statistics
  quantizer.npz    scripts/ckpt_converter/quant_utils.py::pseudo_quantize_tensor (:96-140) on seeded weights

tests/test_oracle_golden_host.py checks oracle/ (and omniserve_amd/ckpt.py) against these files and, when
/root/reference is present, also re-runs this generator in memory and compares it with the committed fixtures.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def reference_available():
    return os.path.isdir(os.path.join(REF, "omniserve"))


def import_reference():
    """Make `import omniserve...` work on CPU: our mirror packages first on sys.path, then the reference checkout."""
    for p in (REF, ROOT):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)          # omniserve_backend / block_sparse_attn / flash_attn = this repo's mirrors
    if not torch.cuda.is_available():
        torch.cuda.current_device = lambda: 0
    import omniserve  # noqa: F401
    return omniserve


# ---- page choice ---------------------------------------------------------------------------------------------------
def gen_page_choice():
    import omniserve.modeling.layers.decoding_attention as da
    cases = []
    rng = np.random.default_rng(2024)
    real_selector = da.fused_attention_selector      # replaced by canned scores below, restored at the end
    for (B, Hq, tpb, sub, budget, timestep, ties) in [
        (2, 4, 64, 16, 256, 700, False),       # 11 pages, budget 4 pages
        (1, 8, 64, 16, 4096, 20000, False),    # 313 pages, budget 64 pages (configs[3] parameters)
        (1, 2, 16, 8, 64, 100, False),         # small pages
        (1, 2, 64, 16, 128, 640, False),       # history ends a page exactly: timestep % tpb == 0
        (1, 2, 64, 16, 256, 700, True),        # exact ties between pages (documents torch.topk's tie order)
        (2, 2, 64, 16, 4096, 1000, False),     # timestep <= budget: every page is taken
    ]:
        w = da.DecodingAttentionWrapper(0, True, 128, None, 1 << 20, tpb, 128, 500000.0, None, True, "fine_grained",
                                        {"INT4_ENABLED": True, "ZEROS_ENABLED": True}, True, 1, sub, budget, 2048, 4)
        subs = tpb // sub
        total_pages = timestep // tpb + 1
        nsub = total_pages * subs
        if ties:
            vals = rng.integers(0, 3, size=(B, Hq, nsub)).astype(np.float16)
        else:   # distinct page maxima: a random permutation of distinct fp16 values per head
            vals = np.stack([rng.permutation(nsub) for _ in range(B * Hq)]).reshape(B, Hq, nsub).astype(np.float16) / 8
        stats = torch.from_numpy(vals.copy())
        da.fused_attention_selector = types.SimpleNamespace(single_query_page_selector=lambda *a, **k: stats)
        q = torch.zeros((B, Hq, 128), dtype=torch.float16)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sel = w.dynamic_select_topk_pages(q, q, q, None, None, None, None, None, 128, 256, 2, 5, 0, 0, 4, 4,
                                              timestep, 512)
        cases.append(dict(B=B, Hq=Hq, tokens_per_block=tpb, sub_chunk=sub, budget=budget, timestep=timestep, ties=ties,
                          stats=vals.astype(np.float32).reshape(-1).tolist(), selected=sel.numpy().tolist(),
                          selected_dtype=str(sel.dtype)))
    da.fused_attention_selector = real_selector
    return cases


# ---- streaming ring of the block manager -------------------------------------------------------------------------------
def gen_ring_map():
    from omniserve.core.block_manager import BaseBlockSpaceManager
    from omniserve.sequence import Sequence, SequenceGroup, SequenceStatus
    out = []
    for (bs, sink_blocks, local_blocks, prompt_len, appended) in [(64, 2, 5, 1000, 300), (64, 1, 3, 100, 400),
                                                                  (16, 1, 4, 40, 100), (64, 2, 5, 64 * 7, 64 * 3)]:
        mgr = BaseBlockSpaceManager(bs, 512, 0, watermark=0.0, sink_local_blocks=(sink_blocks, local_blocks))
        seq = Sequence(0, "", list(range(prompt_len)), bs)
        grp = SequenceGroup("r", [seq], None, 0.0)
        mgr.allocate(grp, ifb_mode=True)
        seq.status = SequenceStatus.RUNNING
        tables = [[b.block_number for b in mgr.block_tables[0]]]
        for t in range(appended):
            seq.append_token_id(7, {7: 0.0})
            mgr.append_slot(seq)
            if (prompt_len + t + 1) % bs in (0, 1):
                tables.append([b.block_number for b in mgr.block_tables[0]])
        out.append(dict(block_size=bs, sink_blocks=sink_blocks, local_blocks=local_blocks, prompt_len=prompt_len,
                        appended=appended, final_table=[b.block_number for b in mgr.block_tables[0]],
                        num_logical_blocks=len(seq.logical_token_blocks)))
    return out


# ---- head classes ------------------------------------------------------------------------------------------------------
def gen_head_masks():
    from omniserve.modeling.layers.ctx_attn.ctx_attn_init import init_ctx_sparse_attn, init_sparse_kv_cache
    out = []
    for (flags_per_layer, Hq, Hk, sink, local, dec_sink, dec_local, tpb) in [
        ([[1, 0, 0, 1], [0, 0, 1, 1]], 8, 4, 128, 8192, 128, 256, 64),
        ([[1, 0, 1, 0, 1, 0, 1, 0]], 32, 8, 64, 1024, 64, 128, 64),
        ([[1, 1, 1, 1]], 8, 4, 128, 8192, 128, 256, 64),            # all-dense layer: head_mask_type stays None
    ]:
        class Attn(torch.nn.Module):
            pass

        class Layer(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.self_attn = Attn()

        class Inner(torch.nn.Module):
            def __init__(self, n):
                super().__init__()
                self.layers = torch.nn.ModuleList([Layer() for _ in range(n)])

        class Model(torch.nn.Module):
            def __init__(self, n):
                super().__init__()
                self.model = Inner(n)
                self.w = torch.nn.Parameter(torch.zeros(1, dtype=torch.float16))
                self.total_num_heads, self.total_num_kv_heads = Hq, Hk

        cfg = types.SimpleNamespace(
            sparse_kv_cache_enabled=lambda: True, sparse_context_enabled=lambda: True,
            get_full_attention_heads=lambda: np.asarray(flags_per_layer), get_ctx_sink_size=lambda: sink,
            get_ctx_local_size=lambda: local, retrieval_head_num=lambda i: int(sum(flags_per_layer[i])),
            streaming_head_num=lambda i: int(len(flags_per_layer[i]) - sum(flags_per_layer[i])),
            get_dec_sink_block_num=lambda: dec_sink // tpb, get_dec_local_block_num=lambda: dec_local // tpb + 1,
            get_dec_sink_size=lambda: dec_sink, get_dec_local_size=lambda: dec_local)
        m = Model(len(flags_per_layer))
        init_ctx_sparse_attn(m, cfg)
        init_sparse_kv_cache(m, cfg)
        layers = []
        for i, layer in enumerate(m.model.layers):
            a = layer.self_attn
            layers.append(dict(
                flags=flags_per_layer[i],
                head_mask_type=None if a.head_mask_type is None else a.head_mask_type.tolist(),
                streaming_info=None if a.streaming_info is None else a.streaming_info.tolist(),
                retrieval_head_flags=a.retrieval_head_flags.tolist(), head_rank_table=a.head_rank_table.tolist(),
                pooling_heads_idx=a.pooling_heads_idx.tolist(), num_retrieval_kv_heads=a.num_retrieval_kv_heads,
                num_streaming_kv_heads=a.num_streaming_kv_heads, sink_blocks=a.sink_blocks, local_blocks=a.local_blocks))
        out.append(dict(Hq=Hq, Hk=Hk, ctx_sink=sink, ctx_local=local, dec_sink=dec_sink, dec_local=dec_local, layers=layers))
    return out


# ---- page sizes ----------------------------------------------------------------------------------------------------------
def gen_page_bytes():
    import omniserve.worker.cache_engine as ce
    out = []
    saved = (ce.BaseCacheEngine.allocate_gpu_cache, ce.BaseCacheEngine.allocate_cpu_cache, torch.cuda.current_stream)
    ce.BaseCacheEngine.allocate_gpu_cache = lambda self: None
    ce.BaseCacheEngine.allocate_cpu_cache = lambda self: None
    torch.cuda.current_stream = lambda *a, **k: object()
    try:
        for (heads, head_size, block, int4, sparse_mode, subs, mode) in [
            (8, 128, 64, True, 0, 4, "retrieval"), (4, 128, 64, True, 1, 4, "retrieval"),
            (4, 128, 64, True, 1, 4, "streaming"), (4, 128, 64, False, 1, 4, "retrieval"),
            (8, 128, 64, False, 0, 4, "retrieval"), (1, 128, 64, True, 0, 2, "retrieval"),
            (2, 128, 16, True, 1, 2, "retrieval"),
        ]:
            sp = types.SimpleNamespace(get_sparse_decode_mode=lambda: sparse_mode,
                                       get_dec_sub_chunk_per_block=lambda: subs)
            model_config = types.SimpleNamespace(get_head_size=lambda: head_size, dtype=torch.float16, sp_attn_config=sp)
            cache_config = types.SimpleNamespace(block_size=block, cache_dtype="int8")
            eng = ce.BaseCacheEngine(heads, 4, 0, object(), None, cache_config, model_config, None,
                                     {"INT4_ENABLED": int4, "ZEROS_ENABLED": int4}, mode)
            out.append(dict(heads=heads, head_size=head_size, block_size=block, int4=int4, sparse_decode_mode=sparse_mode,
                            sub_chunk_per_block=subs, cache_mode=mode, num_bytes_per_block=int(eng.num_bytes_per_block),
                            num_bytes_k_stats_per_block=int(eng.num_bytes_k_stats_per_block)))
    finally:
        ce.BaseCacheEngine.allocate_gpu_cache, ce.BaseCacheEngine.allocate_cpu_cache, torch.cuda.current_stream = saved
    return out


# ---- the converter's fake quantizer ----------------------------------------------------------------------------------------
def gen_quantizer():
    spec = importlib.util.spec_from_file_location("ref_quant_utils", os.path.join(REF, "scripts/ckpt_converter/quant_utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = torch.Generator().manual_seed(77)
    w = torch.randn((48, 256), generator=g) * 0.05
    w[3] = 0.0                                         # constant row: the clamp(min=1e-5) branch
    w[5, :128] *= 30.0                                 # an outlier group
    out = {"w": w.numpy().copy()}
    for tag, (bits, gs) in {"w4_chn": (4, -1), "w4_g128": (4, 128), "w8_chn": (8, -1)}.items():
        dq, scales, zeros = mod.pseudo_quantize_tensor(w.clone(), n_bit=bits, zero_point=True, q_group_size=gs,
                                                       get_scale_zp=True)
        out[tag + "_dq"] = dq.numpy(); out[tag + "_scales"] = scales.numpy(); out[tag + "_zeros"] = zeros.numpy()
    return out


def generate():
    import_reference()
    host = dict(page_choice=gen_page_choice(), ring_map=gen_ring_map(), head_masks=gen_head_masks(),
                page_bytes=gen_page_bytes())
    return host, gen_quantizer()


def main():
    host, quant = generate()
    with open(os.path.join(HERE, "host_logic.json"), "w") as f:
        json.dump(host, f, separators=(",", ":"))
    np.savez_compressed(os.path.join(HERE, "quantizer.npz"), **quant)
    print("written:", os.path.join(HERE, "host_logic.json"), os.path.join(HERE, "quantizer.npz"))


if __name__ == "__main__":
    main()
