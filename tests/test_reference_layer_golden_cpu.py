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
def test_fixture_matches_the_reference_layer_today(golden_dir):
    from tests.golden import make_golden_layer
    fresh = make_golden_layer.generate()
    z = np.load(os.path.join(golden_dir, "decoder_layer_w4a8kv4.npz"))
    assert sorted(fresh) == sorted(z.files)
    for k in z.files:
        assert np.array_equal(np.asarray(fresh[k]).view(np.uint8), z[k].view(np.uint8)), k
