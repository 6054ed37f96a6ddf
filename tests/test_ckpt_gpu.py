"""A (tiny, synthetic, HF-style) QServe checkpoint written by omniserve_amd/ckpt.py loads into the decode runner and the
loaded layers compute what the oracle computes from the checkpoint's own tensors."""
import numpy as np
import pytest
import torch

from omniserve_amd import ckpt
from oracle import w4a8
from tests.util import assert_f16_equal, quantize_act, to_dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("group_size", [-1, 128])
def test_checkpoint_loads_into_decode_runner(tmp_path, group_size):
    from omniserve_amd.runtime import DecodeRunner
    _, cfg = ckpt.make_tiny_checkpoint(str(tmp_path), group_size=group_size, seed=2)
    state = ckpt.load_state_dict(str(tmp_path))
    dev = torch.device("cuda:0")
    r = DecodeRunner(cfg, batch=4, context=70, max_new=8, device=dev, seed=0, use_graph=True, fused=1)
    r.step()                                   # capture first: the loader must keep every device pointer
    ckpt.load_into_runner(r, state)
    L = r.layers[1]
    pre = "model.layers.1.mlp."
    f = ckpt.fused_linear(state, [pre + "gate_proj", pre + "up_proj"], group_size)
    assert torch.equal(L["gate_up"].qweight.cpu(), f["qweight"].view(torch.int8))
    x = np.random.default_rng(0).standard_normal((4, cfg.hidden)).astype(np.float16)
    a, sa, asum = quantize_act(x)
    out = torch.empty((4, 2 * cfg.inter), dtype=torch.float16, device=dev)
    L["gate_up"].forward(to_dev(a), to_dev(sa), to_dev(asum), out)
    torch.cuda.synchronize()
    if group_size == -1:
        want = w4a8.gemm_per_chn(a, f["qweight"].numpy(), f["s1_scales"].numpy(), sa, f["s1_szeros"].numpy(), asum)
    else:
        want = w4a8.gemm_per_group(a, f["qweight"].numpy(), f["s2_zeros"].numpy(), f["s2_scales"].numpy(),
                                   f["s1_scales"].numpy(), sa)
    assert_f16_equal(out, want, "gate_up of the loaded checkpoint")
    assert torch.equal(r.embed.cpu(), state["model.embed_tokens.weight"].half())
    for _ in range(3):
        r.step()                               # graph replays over the loaded weights
    torch.cuda.synchronize()
    assert torch.isfinite(r.x.float()).all()
    assert ((r.tokens >= 0) & (r.tokens < cfg.vocab)).all()
