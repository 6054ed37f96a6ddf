"""tests/golden/decoder_layer_w4a8kv4.npz is what the reference's own LlamaDecoderLayer produces today: where
/root/reference is present (build container) the generator is re-run in memory and compared with the committed file."""
import os

import numpy as np
import pytest

from tests import refstack


def test_fixture_is_complete(golden_dir):
    z = np.load(os.path.join(golden_dir, "decoder_layer_w4a8kv4.npz"))
    hidden, inter, hq, hk, d, tpb, B, L, steps, pages = [int(t) for t in z["shape"]]
    assert z["prefill_in"].shape == (B * L, hidden) and z["prefill_out"].shape == (B * L, hidden)
    assert z["qkv.qweight"].shape == ((hq + 2 * hk) * d, hidden // 2) and z["down.qweight"].shape == (hidden, inter // 2)
    for s in range(steps):
        assert z["decode%d_out" % s].shape == (B, hidden)
        assert z["decode%d_k_pages" % s].shape[:2] == (B, pages)
    # the second decode step appended exactly one more token row than the first (4-bit data + scale + zero of 1 head)
    diff = (z["decode1_k_pages"] != z["decode0_k_pages"]).sum(axis=(1, 2))
    assert (diff > 0).all() and (diff <= d // 2 + 4).all()


@pytest.mark.skipif(not refstack.reference_available(), reason="/root/reference is not on this machine")
@pytest.mark.parametrize("variant", ["base", "h4"])
@pytest.mark.parametrize("group_size", [-1, 128])
def test_fixture_matches_the_reference_layer_today(golden_dir, group_size, variant):
    from tests.golden import make_golden_layer
    fresh = make_golden_layer.generate(group_size, variant)
    z = np.load(make_golden_layer.out_path(group_size, variant))
    assert sorted(fresh) == sorted(z.files)
    for k in z.files:
        assert np.array_equal(np.asarray(fresh[k]).view(np.uint8), z[k].view(np.uint8)), k


# ---- the LServe layer (llama_w8a8_unpad.py, sparse context + dynamic sparse decoding): tests/golden/lserve_layer_{kv8,kv4}.npz
@pytest.mark.parametrize("fmt", ["kv8", "kv4"])
def test_lserve_fixture_is_complete_and_tells_the_refresh_story(golden_dir, fmt):
    z = np.load(os.path.join(golden_dir, "lserve_layer_%s.npz" % fmt))
    (hidden, inter, hq, hk, d, tpb, B, L, steps, rpages, spages, subs, budget, interval, cs, cl) = [int(t) for t in z["shape"]]
    nr, ns, sink, local, sink_blocks, local_blocks = [int(t) for t in z["head_setup"]]
    assert (nr, ns) == (1, 1) and z["retrieval_head_flags"].tolist() == [1, 0]
    assert z["head_mask_type"].tolist() == [0, 0, -1, -1] and z["streaming_info"].tolist() == [cs, cl] * hq
    assert (sink_blocks, local_blocks) == (sink // tpb, local // tpb + 1) and spages == sink_blocks + local_blocks
    row = d if fmt == "kv8" else d // 2
    assert z["prefill_rk"].shape == (B, rpages, nr * tpb * (row + 4) + 2 * subs * nr * d * 2)      # K pages carry statistics
    assert z["prefill_rv"].shape == (B, rpages, nr * tpb * (row + 4)) and z["prefill_sk"].shape == (B, spages, ns * tpb * (row + 4))
    pages = [z["decode%d_pages" % s][0, 0].tolist() for s in range(steps)]
    newest = [(L + s) // tpb for s in range(steps)]            # page of the token generated at step s
    # steps 1, 2 and 4 refresh (nothing cached / even length); step 3 re-uses the cached selection although it crossed into
    # a new page, so its last entry is still the previous page (decoding_attention.py:259-260)
    assert pages[0][-1] == newest[0] and pages[1][-1] == newest[1] and pages[3][-1] == newest[3]
    assert newest[2] == newest[1] + 1 and pages[2] == pages[1]
    assert all(len(p) == max(3, budget // tpb) for p in pages)
    # every step appends one token row to the retrieval pool, and the streaming ring changes too
    for s in range(steps):
        prev = "prefill" if s == 0 else "decode%d" % (s - 1)
        for name in ("rk", "rv", "sk", "sv"):
            assert (z["decode%d_%s" % (s, name)] != z["%s_%s" % (prev, name)]).any(), (s, name)


@pytest.mark.skipif(not refstack.reference_available(), reason="/root/reference is not on this machine")
@pytest.mark.parametrize("fmt", ["kv8", "kv4"])
def test_lserve_fixture_matches_the_reference_layer_today(golden_dir, fmt):
    from tests.golden import make_golden_lserve_layer
    fresh = make_golden_lserve_layer.generate(fmt)
    z = np.load(os.path.join(golden_dir, "lserve_layer_%s.npz" % fmt))
    assert sorted(fresh) == sorted(z.files)
    for k in z.files:
        assert np.array_equal(np.asarray(fresh[k]).view(np.uint8), z[k].view(np.uint8)), k
