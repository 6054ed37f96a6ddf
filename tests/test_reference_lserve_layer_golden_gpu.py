"""The LServe decode driver on the MI355X against vectors produced by the REFERENCE's own LServe layer
(tests/golden/make_golden_lserve_layer.py: `llama_w8a8_unpad.py::LlamaDecoderLayer`, unmodified, with the reference's own
`sparse_attn_init` / `init_sparse_kv_cache` / `init_ctx_sparse_attn`, over the oracle-backed C-ABI): W8A8 linears, one
retrieval and one streaming kv head, statistics pooling, page selector + top-k every second step, sparse attention; KV8
per_tensor and KV4 fine_grained pages.  The runner either starts from the page pools the reference's context stage left
behind, or runs its own context stage (LServeDecodeRunner.prefill) on the reference's prompt -- which must leave exactly
those pools, statistics included -- and then replays the four generation steps with the reference's inputs.  What must hold:
  * the selected pages of every step equal the reference's (the selector scores go through fp16: a different choice would
    need two page scores within 2 ulp of each other -- not the case for these vectors), including WHEN the selection is
    refreshed and that a cached selection is kept as it is across a page boundary (step 3);
  * all four page pools -- appended K / V rows of the retrieval head, its updated min/max statistics, the streaming head's
    ring -- are byte-identical after every step (integer / bit-exact arithmetic up to the cache write);
  * the hidden states agree within the bound of test_reference_layer_golden_gpu.py (attention softmax in fp32 / fp16 on
    the GPU vs f64 in the oracle, re-quantised twice on the way out).
Run for the reference call sequence (eager, fused off) and for the fused entry points the benchmark uses."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load(golden_dir, fmt):
    z = np.load(os.path.join(golden_dir, "lserve_layer_%s.npz" % fmt))
    return {k: z[k] for k in z.files}


def _index(pool, table, which):
    return ((table[:, which, :] - pool.data_ptr()) // pool.shape[1]).long()


def _inject(pool, table, which, pages):
    idx = _index(pool, table, which)[:, : pages.shape[1]]
    pool[idx.reshape(-1)] = torch.from_numpy(pages.reshape(-1, pages.shape[2])).to(pool.device)


def _extract(pool, table, which, n):
    idx = _index(pool, table, which)[:, :n]
    return pool[idx.reshape(-1)].reshape(idx.shape[0], n, pool.shape[1]).cpu().numpy()


def _close(got, want, what):
    got, want = got.astype(np.float32), want.astype(np.float32)
    err = float(np.abs(got - want).max())
    rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    msg = "%s: max |diff| %.4g, max |want| %.4g, rel L2 %.4g" % (what, err, np.abs(want).max(), rel)
    assert rel <= 0.08 and err <= 0.06 * float(np.abs(want).max()), msg


@pytest.mark.parametrize("fused,own_prefill", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("fmt", ["kv8", "kv4"])
def test_lserve_runner_matches_reference_layer_vectors(golden_dir, fmt, fused, own_prefill):
    from omniserve_amd.lserve_runtime import LServeDecodeRunner
    from omniserve_amd.runtime import LlamaConfig
    v = _load(golden_dir, fmt)
    (hidden, inter, hq, hk, d, tpb, B, L, steps, rpages, spages, subs, budget, interval, cs, cl) = [int(t) for t in v["shape"]]
    base, eps, ksc, vsc = [float(t) for t in v["floats"]]
    nr, ns, sink, local, sink_blocks, local_blocks = [int(t) for t in v["head_setup"]]
    dev = torch.device("cuda:0")
    cfg = LlamaConfig(hidden=hidden, inter=inter, heads=hq, kv_heads=hk, head_dim=d, layers=1, vocab=B * steps,
                      rope_theta=base, eps=eps)
    r = LServeDecodeRunner(cfg, batch=B, context=L, max_new=steps + 1, device=dev, seed=3, kv_format=fmt, sink=sink,
                           local=local, budget_tokens=budget, selector_interval=interval, sub_chunk_per_block=subs,
                           use_graph=False, fused=fused, ctx_sink=cs, ctx_local=cl)
    # what the reference's initialisers derived must be what the runner derives
    assert r.head_mask_type.cpu().tolist() == v["head_mask_type"].tolist()
    assert r.streaming_info.cpu().tolist() == v["streaming_info"].tolist()
    assert (r.nr, r.ns, r.sink_blocks, r.local_blocks, r.tpb) == (nr, ns, sink_blocks, local_blocks, tpb)
    assert r.flags.cpu().tolist() == v["retrieval_head_flags"].tolist()
    assert np.allclose(r.kv_qo.cpu().numpy(), [ksc, vsc])
    Ly = r.layers[0]
    for name in ("qkv", "o", "gate_up", "down"):
        Ly[name].weight.copy_(torch.from_numpy(v[name + ".weight"]).to(dev))
        Ly[name].dequant_scale.copy_(torch.from_numpy(v[name + ".dequant_scale"]).to(dev))
    Ly["ln1"].copy_(torch.from_numpy(v["ln1"]).to(dev))
    Ly["ln2"].copy_(torch.from_numpy(v["ln2"]).to(dev))
    r.embed.copy_(torch.from_numpy(np.concatenate([v["decode%d_in" % s] for s in range(steps)], axis=0)).to(dev))
    rk, rv, sk, sv = r.pools[0]
    assert rk.shape[1] == v["prefill_rk"].shape[2] and sk.shape[1] == v["prefill_sk"].shape[2]     # page bytes incl. statistics
    assert r.retr_tables[0].shape[2] >= rpages and r.strm_tables[0].shape[2] == spages
    for pool in (rk, rv, sk, sv):
        pool.zero_()
    pools = (("rk", rk, r.retr_tables[0], 0), ("rv", rv, r.retr_tables[0], 1), ("sk", sk, r.strm_tables[0], 0),
             ("sv", sv, r.strm_tables[0], 1))
    if own_prefill:      # the runner's context stage on the reference's prompt (two MLP chunks, as chunk_prefill_size would cut)
        x = r.prefill(hidden=torch.from_numpy(v["prefill_in"]), chunk=400)
        torch.cuda.synchronize()
        for name, pool, tab, which in pools:
            want = v["prefill_" + name]
            assert np.array_equal(_extract(pool, tab, which, want.shape[1]), want), "context stage: %s pages differ" % name
        _close(x.cpu().numpy(), v["prefill_out"], "context stage hidden state")
        assert int(r.lengths[0]) == L and r.page_idx[0].shape[2] == max(3, budget // tpb)
    else:                # the state the reference's context stage left behind
        for name, pool, tab, which in pools:
            _inject(pool, tab, which, v["prefill_" + name])
    for s in range(steps):
        r.tokens.copy_(torch.arange(s * B, (s + 1) * B, device=dev))
        r.step()
        torch.cuda.synchronize()
        tag = "decode%d" % s
        assert int(r.lengths[0]) == L + s + 1
        # q heads of the retrieval kv head(s); a streaming head's scores are all zero, its (unused) entries are whatever
        # order topk leaves among equals
        rq = np.repeat(v["retrieval_head_flags"], hq // hk).astype(bool)
        assert np.array_equal(r.page_idx[0].cpu().numpy()[:, rq], v[tag + "_pages"][:, rq]), \
            "%s: selected pages %s, reference %s" % (tag, r.page_idx[0][:, rq].tolist(), v[tag + "_pages"][:, rq].tolist())
        for name, pool, tab, which in pools:
            want = v["%s_%s" % (tag, name)]
            assert np.array_equal(_extract(pool, tab, which, want.shape[1]), want), "%s: %s pages differ" % (tag, name)
        _close(r.x.cpu().numpy(), v[tag + "_out"], tag + " hidden state")
