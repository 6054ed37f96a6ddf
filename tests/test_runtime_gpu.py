"""The decode runner must produce the same tokens / residual stream whichever fusion level is used
(the fused entry points are bit-identical to the reference call sequence) and with or without the
HIP graph."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(fused, graph, steps=3):
    from omniserve_amd.runtime import DecodeRunner, LlamaConfig
    dev = torch.device("cuda:0")
    r = DecodeRunner(LlamaConfig.tiny(), batch=5, context=70, max_new=8, device=dev, seed=7, use_graph=graph, fused=fused)
    toks = []
    for _ in range(steps):
        r.step()
        toks.append(r.tokens.clone())
    torch.cuda.synchronize()
    return torch.stack(toks).cpu(), r.x.clone().cpu(), [p.clone().cpu() for p in r.pools[0]]


def test_fusion_levels_and_graph_agree_bitwise():
    ref_t, ref_x, ref_p = _run(0, False)
    for fused, graph in [(1, False), (2, False), (2, True)]:
        t, x, pools = _run(fused, graph)
        assert torch.equal(t, ref_t), (fused, graph)
        assert torch.equal(x.view(torch.int16), ref_x.view(torch.int16)), (fused, graph)
        for a, b in zip(pools, ref_p):
            assert torch.equal(a, b)
