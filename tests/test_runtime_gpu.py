"""The decode runner must produce the same tokens / residual stream whichever fusion level is used
(the fused entry points are bit-identical to the reference call sequence) and with or without the
HIP graph."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(fused, graph, steps=3, group_size=-1, batch=5):
    from omniserve_amd.runtime import DecodeRunner, LlamaConfig
    dev = torch.device("cuda:0")
    cfg = LlamaConfig.tiny()
    cfg.group_size = group_size
    r = DecodeRunner(cfg, batch=batch, context=70, max_new=8, device=dev, seed=7, use_graph=graph, fused=fused)
    assert r.fused == (min(int(fused), 3) if batch <= 16 else min(int(fused), 2))     # level 3 needs the 16-row GEMV tile
    assert r.pairs == (int(fused) >= 4 and batch <= 16)     # level 4 = level 3 + the (norm -> GEMV) pairs as single launches
    toks = []
    for _ in range(steps):
        r.step()
        toks.append(r.tokens.clone())
    torch.cuda.synchronize()
    return torch.stack(toks).cpu(), r.x.clone().cpu(), [p.clone().cpu() for p in r.pools[0]]


@pytest.mark.parametrize("group_size,batch", [(-1, 5), (128, 5), (128, 40), (-1, 160), (128, 130)])
def test_fusion_levels_and_graph_agree_bitwise(group_size, batch):
    """(g128 at level 2 = the per-group partial GEMM + the slab-consuming norm; batch 40 = the 64-row GEMV tile; level 3 =
    no quantiser row kernels: SiLU in the gate_up epilogue, o / down quantising on the fly -- it falls back to 2 at batch 40;
    batch 130 / 160 = the 128 x 256 tile with K slices over grid.y: slab-only form for o / down, split + slab epilogue elsewhere;
    level 4 = level 3 with (add + norm + quant) -> qkv and -> gate_up as single launches, csrc/norm_gemv_fused.h)"""
    ref_t, ref_x, ref_p = _run(0, False, group_size=group_size, batch=batch)
    for fused, graph in [(1, False), (2, False), (2, True), (3, False), (3, True), (4, False), (4, True)]:
        t, x, pools = _run(fused, graph, group_size=group_size, batch=batch)
        assert torch.equal(t, ref_t), (fused, graph)
        assert torch.equal(x.view(torch.int16), ref_x.view(torch.int16)), (fused, graph)
        for a, b in zip(pools, ref_p):
            assert torch.equal(a, b)


def test_hidden_5120_layer_takes_level_2_and_steps():
    """A Llama-2-13B-shaped layer (hidden 5120: the gate_up GEMV's plan splits K across workgroups, which the SiLU-epilogue
    form of level 3 does not take) asked for the default level: the runner must settle on level 2 up front and decode --
    bit-identically to the reference call sequence -- instead of raising in its first step."""
    from omniserve_amd.runtime import DecodeRunner, LlamaConfig
    dev = torch.device("cuda:0")
    cfg = LlamaConfig(hidden=5120, inter=13824, heads=40, kv_heads=40, layers=1, vocab=512)
    out = []
    for fused in (True, 0):
        r = DecodeRunner(cfg, batch=16, context=70, max_new=8, device=dev, seed=5, use_graph=False, fused=fused)
        assert r.fused == (2 if fused else 0)
        for _ in range(2):
            r.step()
        torch.cuda.synchronize()
        out.append((r.tokens.clone().cpu(), r.x.clone().cpu()))
        del r
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1].view(torch.int16), out[1][1].view(torch.int16))


def test_prefill_is_deterministic_and_decodable():
    """Two prefill calls on the same prompt write bit-identical KV pages and pick the same token; the pages are then
    readable by the decode path (finite activations, valid tokens)."""
    from omniserve_amd.runtime import DecodeRunner, LlamaConfig
    cfg = LlamaConfig.tiny()
    r = DecodeRunner(cfg, 3, 70, 8, torch.device("cuda:0"), seed=11, use_graph=False, fused=2)
    gen_state = r.gen.get_state()
    r.prefill(70)
    pools_a = [[p.clone() for p in layer] for layer in r.pools]
    tok_a = r.tokens.clone()
    r.gen.set_state(gen_state)
    r.prefill(70)
    torch.cuda.synchronize()
    assert torch.equal(tok_a, r.tokens)
    for la, lb in zip(pools_a, r.pools):
        for pa, pb in zip(la, lb):
            assert torch.equal(pa, pb)
    assert int(r.lengths[0]) == 70
    for _ in range(3):
        r.step()
    torch.cuda.synchronize()
    assert torch.isfinite(r.x.float()).all()
    assert int(r.lengths[0]) == 73
    assert ((r.tokens >= 0) & (r.tokens < cfg.vocab)).all()


def _kv_rows(runner, layer, L):
    """(K rows, V rows) of the first L tokens of every sequence of one layer, gathered through the block tables:
    uint8 [B, Hkv, L, 64 + 4] = packed codes | fp16 scale | fp16 zero of each token row."""
    kl, tpb, B = runner.kl, runner.tpb, runner.B
    data_bytes = kl * tpb * 64
    out = []
    for kv in range(2):
        pool = runner.pools[layer][kv]
        base = pool.data_ptr()
        idx = ((runner.block_tables[layer][:, kv] - base) // runner.page_bytes).cpu()      # [B, pages]
        pool_c = pool.cpu()
        rows = torch.empty((B, kl, L, 68), dtype=torch.uint8)
        for b in range(B):
            for t in range(L):
                page = pool_c[int(idx[b, t // tpb])]
                for h in range(kl):
                    o = (h * tpb + t % tpb)
                    rows[b, h, t, :64] = page[o * 64:(o + 1) * 64]
                    tail = data_bytes + 2 * o
                    rows[b, h, t, 64:66] = page[tail:tail + 2]
                    rows[b, h, t, 66:68] = page[tail + 2 * kl * tpb:tail + 2 * kl * tpb + 2]
        out.append(rows)
    return out


@pytest.mark.parametrize("fused", [0, 2])
def test_prefill_matches_token_by_token_decode(fused):
    """The context stage (prefill GEMMs at M = B*L, in-place RoPE + KV4 page writer, varlen causal attention) against the
    generation stage fed the same prompt one token at a time from an EMPTY cache (decode GEMVs, fused RoPE + append,
    paged KV4 decode attention).  Layer 0's K/V rows depend on the token embeddings only and go through bit-exact
    kernels on both paths: they must be byte-identical.  Deeper state differs by what the two attention kernels may
    differ by: prefill attends fp16 K/V, decode the KV4-quantised cache (quantisation noise of 4-bit K/V, as upstream),
    so the final hidden state is compared with a tolerance that is measured, not assumed: <= 0.25 of its largest entry (see the assertion message for the measured value)."""
    from omniserve_amd.runtime import DecodeRunner, LlamaConfig
    dev = torch.device("cuda:0")
    cfg = LlamaConfig.tiny()
    B, L = 3, 70      # crosses the 64-token page boundary
    a = DecodeRunner(cfg, B, L, 8, dev, seed=21, use_graph=False, fused=fused)
    gen_state = a.gen.get_state()
    prompt = torch.randint(0, cfg.vocab, (B * L,), device=dev, generator=a.gen).view(B, L)
    a.gen.set_state(gen_state)       # prefill() draws the same B*L tokens from the runner's generator
    a.prefill(L)
    torch.cuda.synchronize()
    x_prefill = a.x.clone().float().cpu()
    rows_a = _kv_rows(a, 0, L)

    b = DecodeRunner(cfg, B, 0, L + 8, dev, seed=21, use_graph=False, fused=fused)    # same weights, empty cache
    for layer in b.pools:            # the synthetic runner pre-fills its pools with random pages: start from zeros
        for pool in layer:
            pool.zero_()
    # same page order as runner a is not needed: rows are gathered through each runner's own tables
    for t in range(L):
        b.tokens.copy_(prompt[:, t])
        b.step()
    torch.cuda.synchronize()
    assert int(b.lengths[0]) == L
    rows_b = _kv_rows(b, 0, L)
    for ra, rb, name in zip(rows_a, rows_b, ("K", "V")):
        assert torch.equal(ra, rb), "layer-0 %s rows differ between the prefill writer and the decode append" % name
    x_decode = b.x.float().cpu()
    scale = x_prefill.abs().max().item()
    err = (x_prefill - x_decode).abs().max().item()
    assert err <= 0.25 * scale, "final hidden state: max |prefill - decode| = %g (largest entry %g)" % (err, scale)
