"""The LServe decode driver (omniserve_amd/lserve_runtime.py: W8A8 linears, page selector + top-k, sparse attention over
retrieval / streaming heads; KV8 per_tensor and KV4 fine_grained pages) runs, stays finite, is deterministic, and its
HIP-graph replay with the fused entry points reproduces the eager reference call sequence token for token."""
import pytest
import torch

from omniserve_amd.lserve_runtime import LServeDecodeRunner
from omniserve_amd.runtime import LlamaConfig

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kv_format", ["kv8", "kv4"])
def test_lserve_decode_graph_matches_eager(kv_format):
    dev = torch.device("cuda:0")
    cfg = LlamaConfig.tiny()
    toks, hidden, pools = [], [], []
    # reference call sequence eagerly | fused entry points with the quantiser row kernels (level 2) in a graph | the
    # row-kernel-free layer (level 3: wide merge + row maxima, SiLU epilogue, o / down quantising on the fly) in a graph
    for use_graph, fused in ((False, False), (True, 2), (True, True)):
        r = LServeDecodeRunner(cfg, batch=2, context=700, max_new=16, device=dev, seed=11, kv_format=kv_format,
                               sink=64, local=128, budget_tokens=256, selector_interval=4, use_graph=use_graph,
                               fused=fused)
        seq = []
        for _ in range(9):          # crosses a page boundary (704) and three selector refreshes
            r.step()
            seq.append(r.tokens.clone())
        torch.cuda.synchronize()
        assert torch.isfinite(r.x.float()).all()
        assert int(r.lengths[0]) == 709
        # the newest page is always the last selected entry
        assert int(r.page_idx[0][0, 0, -1]) == 708 // 64
        assert r.rowfree == (fused is True)
        toks.append(torch.stack(seq).cpu())
        hidden.append(r.x.view(torch.int16).cpu())
        pools.append([p.cpu() for layer in r.pools for p in layer])
    for i in (1, 2):
        assert torch.equal(toks[0], toks[i])
        assert torch.equal(hidden[0], hidden[i]), "residual stream differs (bitwise) from the reference call sequence"
        for a, b in zip(pools[0], pools[i]):
            assert torch.equal(a, b), "KV pages differ from the reference call sequence"
