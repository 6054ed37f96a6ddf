"""Parity of the per-tensor KV8 family (fused_attention_per_tensor_{dense,sparse}, LServe's
published `w8a8kv8 per_tensor` configuration) against oracle/kv8.py: prefill writer bit-exact (int8 pages, the
per-row absmax/127 left in the tail, rings of the streaming heads), decode attention within 1e-3 relative of
the f64 oracle with bit-exact appended rows and page statistics, context pooling on KV8 pages bit-exact, page
selection within 2 fp16 ulp (as on KV4 pages)."""
import numpy as np
import pytest
import torch

from oracle import kv4, kv8
from tests.test_fine_grained_gpu import D, FLAGS_MIXED, ROPE_BASE, Case
from tests.util import f16_ulp_diff, to_dev

pytestmark = pytest.mark.gpu

SCALES = (0.03, 0.035)      # N(0,1) data: |x| > 3.81 saturates the K codes, a few V codes saturate too


def test_kv8_codes_oracle_roundtrip():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((64, D)).astype(np.float16)
    oq = np.float32(1.0) / np.float32(0.03)
    c = kv8.kv8_quantize(x, oq)
    assert c.min() == -128 or c.max() == 127                      # saturation is exercised
    y = kv8.kv8_dequant(c, 0.03)
    inside = np.abs(x.astype(np.float32)) < 3.8
    assert np.abs(y - x.astype(np.float32))[inside].max() <= 0.5 * 0.03 + 2e-3


@pytest.mark.parametrize("seq_lens,Hq,flags,tpb,sink,local", [
    ([5, 40, 130, 200], 8, FLAGS_MIXED, 16, 16, 48),
    ([700, 64, 383, 385], 8, FLAGS_MIXED, 64, 128, 256),
    ([100, 300], 16, [0, 0], 64, 64, 128),
    ([90, 33], 4, [1, 1, 1, 1], 16, 16, 32),
])
def test_kv8_prefill_write(seq_lens, Hq, flags, tpb, sink, local):
    Case(seq_lens, Hq, flags, tpb, sink, local, seed=sum(seq_lens), kv8_scales=SCALES).prefill()


def test_kv8_prefill_write_linear_rope_scaling():
    Case([70, 3, 150], 8, FLAGS_MIXED, 16, 16, 48, seed=5, scale=4.0, kv8_scales=SCALES).prefill()


@pytest.mark.parametrize("seq_lens,Hq,flags,tpb,sink,local,steps", [
    ([5, 40, 130, 200], 8, FLAGS_MIXED, 16, 16, 48, 3),
    ([62, 63, 64, 79], 8, FLAGS_MIXED, 16, 16, 48, 4),
    ([700, 64, 383, 385], 16, FLAGS_MIXED, 64, 128, 256, 2),
    ([100, 300], 16, [0, 0], 64, 64, 128, 2),
    ([1500, 1030], 32, [1, 0, 0, 0, 1, 0, 0, 1], 64, 128, 256, 1),
    ([1, 2], 8, [1, 1], 64, 128, 256, 3),                      # (almost) empty history
])
def test_kv8_decode_dense(seq_lens, Hq, flags, tpb, sink, local, steps):
    c = Case(seq_lens, Hq, flags, tpb, sink, local, seed=sum(seq_lens) + Hq, kv8_scales=SCALES)
    c.prefill()
    c.decode(steps)


def test_kv8_decode_dense_linear_rope_scaling():
    c = Case([70, 150], 8, FLAGS_MIXED, 16, 16, 48, seed=9, scale=2.0, kv8_scales=(0.05, 0.02))
    c.prefill()
    c.decode(2)


def _pool_prompt_stats(c, tpb, sub):
    """paged_min_max_pool of the prompt through the mirror module, checked bit-exactly against the oracle."""
    import omniserve_backend.fused_attention_ctx_pool as cp
    cu = np.concatenate([[0], np.cumsum(c.seq_lens)]).astype(np.int32)
    k_post = np.ascontiguousarray(c.qkv_post[:, c.Hq * D:(c.Hq + c.Hk) * D].reshape(-1, c.Hk, D))
    heads = [h for h in range(c.Hk) if c.flags[h]]
    kv4.paged_min_max_pool(k_post, cu, heads, c.rk.pool, c.rk_idx, tpb, sub, row_bytes=c.row)
    cp.paged_min_max_pool(to_dev(k_post), c.g_retr.table, to_dev(cu), to_dev(np.asarray(heads, np.int32)),
                          max(c.seq_lens), sub, tpb, c.nr * c.row, True)
    torch.cuda.synchronize()
    c.check_pools("context pooling")


@pytest.mark.parametrize("seq_lens,Hq,flags,tpb,sink,local,P,sub,steps", [
    ([130, 200, 97], 8, FLAGS_MIXED, 16, 16, 48, 4, 8, 3),
    ([700, 640], 16, FLAGS_MIXED, 64, 128, 256, 6, 16, 2),
    ([255, 256], 8, [1, 1], 64, 128, 256, 3, 32, 2),
])
def test_kv8_decode_sparse_with_pool_and_selector(seq_lens, Hq, flags, tpb, sink, local, P, sub, steps):
    import omniserve_backend.fused_attention_selector as sel
    c = Case(seq_lens, Hq, flags, tpb, sink, local, seed=sum(seq_lens) + P, sub_chunk=sub, kv8_scales=SCALES)
    c.prefill()
    _pool_prompt_stats(c, tpb, sub)

    # page selector on KV8 pages (statistics sit behind 128-B rows)
    lens = np.asarray(c.seq_lens, np.int32) + 1
    q = c.rng.standard_normal((c.B, c.Hq, D)).astype(np.float16)
    want = kv4.page_selector(q, lens, c.flags, c.rank, c.rk.pool, c.rk_idx, c.Hk, c.nr, tpb, sub, ROPE_BASE,
                             row_bytes=c.row)
    kdummy = to_dev(np.zeros((c.B, c.Hk, D), np.float16))
    got = sel.single_query_page_selector(
        to_dev(q), kdummy, kdummy, c.g_retr.table, c.g_strm.table, c.flags_d, c.rank_d, None, to_dev(lens), None,
        65536, tpb, c.nr * c.row, c.ns * c.row, sink, local, c.sink_blocks, c.local_blocks, c.nr, c.ns,
        int(lens.max()) - 1, D, ROPE_BASE, 1.0, True, False, True, sub, c.nr * D, 1000000)
    torch.cuda.synchronize()
    assert tuple(got.shape) == want.shape and f16_ulp_diff(got, want) <= 2, "page selector scores on KV8 pages"

    def dyn_fn(hist):
        dyn = np.zeros((c.B, c.Hq, P), np.int32)
        for b in range(c.B):
            last = (int(hist[b]) - 1) // tpb
            for h in range(c.Hq):
                pick = c.rng.choice(last, size=P - 1, replace=False) if last >= P - 1 else np.arange(P - 1) % max(last, 1)
                dyn[b, h, : P - 1] = np.sort(pick)
                dyn[b, h, P - 1] = last
        return dyn

    c.decode(steps, dyn_fn)


def test_kv8_rejects_unsupported_formats():
    import omniserve_backend.fused_attention_per_tensor_dense as fpd
    c = Case([20], 8, [1, 1], 16, 16, 32, seed=3, kv8_scales=SCALES)
    qkv = to_dev(np.zeros((20, (8 + 4) * D), np.float16))
    lens = to_dev(np.asarray([20], np.int32))
    pad = to_dev(np.zeros((20,), np.int32))
    args = (lens, lens, pad, c.g_retr.table, c.g_strm.table, c.flags_d, c.rank_d, 8, 2, 20, 16, 2 * D, 0, 16, 32, 1, 3,
            2, 0, D, ROPE_BASE, 1.0, 1 << 20, True)
    with pytest.raises(NotImplementedError):     # per_tensor int4 is not implemented
        fpd.apply_bias_rope_update_kv_cache(qkv, c.oq_d, *args, True, False)
    with pytest.raises(RuntimeError):            # scales must be fp32 [2]
        fpd.apply_bias_rope_update_kv_cache(qkv, c.oq_d.half(), *args, False, False)
