"""The runner's decoder layer on the MI355X against vectors produced by the REFERENCE's own model code
(tests/golden/make_golden_layer.py: `llama_w4a8_unpad.py::LlamaDecoderLayer`, unmodified, over the oracle-backed C-ABI):
same packed weights (the reference's packer), same inputs, one context-stage pass over two 70-token prompts and two
generation-stage steps.  Pins the WIRING of omniserve_amd.runtime.DecodeRunner -- call order, buffers, residual handling,
in-place RoPE, cache append, lengths, and the fused entry points of levels 1 / 2 / 3 -- to the reference's layer:
  * the KV4 pages (codes, scales, zeros of every written token) must be byte-identical: everything up to and including the
    cache write is integer / bit-exact arithmetic (norm + quant -> W4A8 GEMM -> RoPE -> 4-bit quantiser);
  * the hidden states go through the attention softmax (HIP: fp32 online softmax, fp16 probabilities; oracle: f64), whose
    fp16 output is re-quantised to int8 twice more on the way out.  This tiny random-weight layer amplifies that: perturbing
    the ORACLE's own attention output by 2e-4 relative (half an fp16 ulp) already moves the layer output by 2-3 % in relative
    L2 with < 50 % of the elements bit-identical (measured with the generator).  Measured HIP vs vectors:
    1.7-5 % -- so the bound here (8 % relative L2, 6 % of the largest value) separates a wiring error (missing residual,
    wrong buffer, wrong order: relative error ~ 1) from rounding, nothing finer; the fine-grained evidence for the
    arithmetic is the per-kernel bit-exact suite."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load(golden_dir, group_size, variant="base"):
    tag = ("" if group_size == -1 else "_g%d" % group_size) + ("" if variant == "base" else "_" + variant)
    z = np.load(os.path.join(golden_dir, "decoder_layer_w4a8kv4%s.npz" % tag))
    return {k: z[k] for k in z.files}


def _pages(r, kv, pages):
    """[B, pages, page bytes] of layer 0 in logical page order, from the runner's raw-pointer block table."""
    pool = r.pools[0][kv]
    tab = r.block_tables[0][:, kv, :pages]
    idx = ((tab - pool.data_ptr()) // r.page_bytes).long()
    return pool[idx.reshape(-1)].reshape(tab.shape[0], pages, r.page_bytes).cpu().numpy()


def _close(got, want, what, rel_bar=0.08, max_bar=0.06):
    got, want = got.astype(np.float32), want.astype(np.float32)
    same = float((got == want).mean())
    err = float(np.abs(got - want).max())
    rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    msg = "%s: max |diff| %.4g, max |want| %.4g, rel L2 %.4g, bit-identical %.1f %%" % (what, err, np.abs(want).max(), rel, 100 * same)
    assert rel <= rel_bar and err <= max_bar * float(np.abs(want).max()), msg
    return msg


# "h4" (four q heads, hidden 512): the smallest layer fusion level 3 -- the headline's level: no quantiser row kernels, SiLU in
# the gate_up epilogue, o / down quantising on the fly -- accepts; the level the runner really ran is asserted below, so a
# silent drop to a lower level cannot pass for it
@pytest.mark.parametrize("variant,fused", [("base", 0), ("base", 1), ("base", 2), ("h4", 0), ("h4", 2), ("h4", 3)])
@pytest.mark.parametrize("group_size", [-1, 128])      # per channel (configs[1]) and g128 (configs[2])
def test_runner_layer_matches_reference_layer_vectors(golden_dir, group_size, variant, fused):
    from omniserve_amd.runtime import DecodeRunner, LlamaConfig
    v = _load(golden_dir, group_size, variant)
    assert int(v["group_size"][0]) == group_size
    hidden, inter, hq, hk, d, tpb, B, L, steps, pages = [int(t) for t in v["shape"]]
    base, eps = [float(t) for t in v["rope_base_eps"]]
    T = B * L
    dev = torch.device("cuda:0")
    cfg = LlamaConfig(hidden=hidden, inter=inter, heads=hq, kv_heads=hk, head_dim=d, layers=1, vocab=T + steps * B,
                      rope_theta=base, eps=eps, group_size=group_size)
    r = DecodeRunner(cfg, B, L, 8, dev, seed=1, use_graph=False, fused=fused)
    assert r.tpb == tpb
    assert r.fused == fused, "the runner dropped fusion level %d to %d on this layer" % (fused, r.fused)
    Ly = r.layers[0]
    for name in ("qkv", "o", "gate_up", "down"):
        for buf in (("qweight", "s1_scales", "s1_szeros") if group_size == -1 else ("qweight", "s1_scales", "s2_scales", "s2_zeros")):
            dst = getattr(Ly[name], buf)
            src = torch.from_numpy(v["%s.%s" % (name, buf)])
            assert tuple(dst.shape) == tuple(src.shape), (name, buf)
            dst.copy_(src.to(dev).view(dst.dtype) if src.dtype != dst.dtype and src.element_size() == dst.element_size()
                      else src.to(dev))
    Ly["ln1"].copy_(torch.from_numpy(v["ln1"]).to(dev))
    Ly["ln2"].copy_(torch.from_numpy(v["ln2"]).to(dev))
    rows = np.concatenate([v["prefill_in"]] + [v["decode%d_in" % s] for s in range(steps)], axis=0)
    r.embed.copy_(torch.from_numpy(rows).to(dev))
    for pool in r.pools[0]:
        pool.zero_()                                  # the reference run started from zeroed pools
    # ---- context stage
    r.prefill(L, tokens=torch.arange(T, device=dev))
    torch.cuda.synchronize()
    assert np.array_equal(_pages(r, 0, pages), v["prefill_k_pages"]), "K pages after prefill"
    assert np.array_equal(_pages(r, 1, pages), v["prefill_v_pages"]), "V pages after prefill"
    # (the wider h4 layer re-quantises twice as many channels behind the attention: measured 6 % relative L2 / 8 % of the largest
    #  value at level 0, i.e. the reference call sequence itself -- the bars still separate rounding from a wiring error, ~ 1)
    bars = dict(rel_bar=0.12, max_bar=0.15) if variant == "h4" else {}
    _close(r._prefill_bufs["x"].cpu().numpy(), v["prefill_out"], "prefill hidden state", **bars)
    # ---- generation stage
    for s in range(steps):
        r.tokens.copy_(torch.arange(T + s * B, T + (s + 1) * B, device=dev))
        r.step()
        torch.cuda.synchronize()
        assert np.array_equal(_pages(r, 0, pages), v["decode%d_k_pages" % s]), "K pages after decode step %d" % s
        assert np.array_equal(_pages(r, 1, pages), v["decode%d_v_pages" % s]), "V pages after decode step %d" % s
        _close(r.x.cpu().numpy(), v["decode%d_out" % s], "decode step %d hidden state" % s, **bars)
