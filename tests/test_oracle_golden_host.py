"""Pin the host-side pieces of the oracle (and of the product's own host logic) to fixtures produced by running the
REFERENCE's Python (tests/golden/make_golden_host.py): page choice, streaming ring of the block manager, head classes,
page sizes, the converter's fake quantizer.  When /root/reference is present (build container) the generator is also
re-run in memory and compared with the committed fixtures, so a stale fixture cannot hide."""
import json
import os

import numpy as np
import pytest

from oracle import kv4, kv8


@pytest.fixture(scope="module")
def host(golden_dir):
    with open(os.path.join(golden_dir, "host_logic.json")) as f:
        return json.load(f)


def test_page_choice_matches_reference(host):
    for c in host["page_choice"]:
        stats = np.asarray(c["stats"], np.float32).astype(np.float16).reshape(c["B"], c["Hq"], -1)
        got = kv4.select_topk_pages(stats, c["tokens_per_block"], c["sub_chunk"], c["budget"], c["timestep"])
        want = np.asarray(c["selected"], np.int32)
        assert c["selected_dtype"] == "torch.int32"
        assert got.shape == want.shape, c["timestep"]
        # the newest page (the page the current token goes to, timestep // tokens_per_block) is always last
        assert (want[..., -1] == c["timestep"] // c["tokens_per_block"]).all()
        if not c["ties"]:
            assert np.array_equal(got, want), (c["timestep"], c["budget"])
        else:   # torch.topk's order among equal scores is unspecified: same scores in the same order, same page SET
            subs = c["tokens_per_block"] // c["sub_chunk"]
            ps = stats.reshape(c["B"], c["Hq"], -1, subs).max(-1)
            assert np.array_equal(np.take_along_axis(ps, got.astype(np.int64), -1),
                                  np.take_along_axis(ps, want.astype(np.int64), -1))
            assert np.array_equal(got[..., -1], want[..., -1])


def test_streaming_ring_matches_block_manager(host):
    for r in host["ring_map"]:
        table = r["final_table"]
        assert len(table) == r["num_logical_blocks"]
        for i, phys in enumerate(table):
            assert table[kv4.ring_block(i, r["sink_blocks"], r["local_blocks"])] == phys
        # the ring never holds more than sink + local distinct pages
        assert len(set(table)) == min(len(table), r["sink_blocks"] + r["local_blocks"])


def test_head_classes_match_ctx_attn_init(host):
    from omniserve_amd.lserve_runtime import head_rank_table
    for m in host["head_masks"]:
        for layer in m["layers"]:
            got = kv4.head_classes(layer["flags"], m["Hq"])
            if layer["head_mask_type"] is None:
                assert got["head_mask_type"] is None
            else:
                assert got["head_mask_type"].tolist() == layer["head_mask_type"]
                assert layer["streaming_info"] == [m["ctx_sink"], m["ctx_local"]] * m["Hq"]
            assert got["retrieval_head_flags"].tolist() == layer["retrieval_head_flags"]
            assert got["head_rank_table"].tolist() == layer["head_rank_table"]
            assert got["pooling_heads_idx"].tolist() == layer["pooling_heads_idx"]
            assert head_rank_table(layer["flags"]) == layer["head_rank_table"]       # the product's own helper
            assert layer["sink_blocks"] == m["dec_sink"] // 64 and layer["local_blocks"] == m["dec_local"] // 64 + 1


def test_page_bytes_match_cache_engine(host):
    for p in host["page_bytes"]:
        heads, d, tpb = p["heads"], p["head_size"], p["block_size"]
        want = p["num_bytes_per_block"] + p["num_bytes_k_stats_per_block"]
        sub = tpb // p["sub_chunk_per_block"]
        row = d // 2 if p["int4"] else d
        plain = kv4.page_bytes(heads, d, tpb) if p["int4"] else kv8.page_bytes(heads, d, tpb)
        assert plain == p["num_bytes_per_block"]
        if p["num_bytes_k_stats_per_block"]:
            assert kv4.stats_page_bytes(heads, d, tpb, sub, row_bytes=row) == want
        else:
            assert plain == want


def test_fake_quantizer_matches_reference(golden_dir):
    from omniserve_amd import ckpt
    import torch
    g = np.load(os.path.join(golden_dir, "quantizer.npz"))
    w = torch.from_numpy(g["w"])
    for tag, (bits, gs) in {"w4_chn": (4, -1), "w4_g128": (4, 128), "w8_chn": (8, -1)}.items():
        dq, scales, zeros = ckpt.pseudo_quantize_tensor(w.clone(), n_bit=bits, q_group_size=gs)
        assert np.array_equal(dq.numpy(), g[tag + "_dq"]), tag
        assert np.array_equal(scales.numpy(), g[tag + "_scales"]), tag
        assert np.array_equal(zeros.numpy(), g[tag + "_zeros"]), tag


def test_fixtures_are_current_when_reference_is_present(host, golden_dir):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_host", os.path.join(golden_dir, "make_golden_host.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not mod.reference_available():
        pytest.skip("/root/reference is not on this machine (GPU box): fixtures are checked as committed")
    fresh_host, fresh_q = mod.generate()
    assert json.loads(json.dumps(fresh_host)) == host
    g = np.load(os.path.join(golden_dir, "quantizer.npz"))
    for k, v in fresh_q.items():
        assert np.array_equal(np.asarray(v), g[k]), k
