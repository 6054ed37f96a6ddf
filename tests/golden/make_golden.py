"""Generate golden fixtures by running the REFERENCE's own Python packer.

Run in the build container only (needs /root/reference, which does not exist
on the GPU box):  python tests/golden/make_golden.py

The reference module omniserve/modeling/layers/quantized_linear/w4a8_linear.py
imports CUDA extension modules at import time and evaluates
torch.cuda.current_device() in a default argument; both are stubbed here so the
pure-torch packing code in ``W4A8OF16LinearDynamicInputScale.from_linear``
(w4a8_linear.py:141-337) runs unmodified on CPU.  Its outputs are stored as the
fixtures ``w4a8_pack_per_chn.npz`` / ``w4a8_pack_per_group.npz`` that
tests/test_oracle_golden.py checks the oracle's closed-form packer against.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/omniserve/modeling/layers/quantized_linear/w4a8_linear.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_packer():
    pkg = types.ModuleType("omniserve_backend")
    pkg.__path__ = []
    sys.modules["omniserve_backend"] = pkg
    for name in ("qgemm_w4a8_per_chn", "qgemm_w4a8_per_group"):
        m = types.ModuleType("omniserve_backend." + name)
        sys.modules["omniserve_backend." + name] = m
        setattr(pkg, name, m)
    torch.cuda.current_device = lambda: "cpu"       # default-arg evaluation at class creation
    torch.Tensor.cuda = lambda self, *a, **k: self   # from_linear calls weight.cuda()
    spec = importlib.util.spec_from_file_location("ref_w4a8_linear", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.W4A8OF16LinearDynamicInputScale


def main():
    cls = load_reference_packer()
    rng = np.random.default_rng(1234)

    # ---- per-channel: N=64, K=128 -------------------------------------------------
    N, K = 64, 128
    u = rng.integers(0, 16, size=(N, K)).astype(np.int64)
    zeros = rng.integers(0, 16, size=(N,)).astype(np.int64)
    s1 = rng.uniform(0.002, 0.02, size=(N,)).astype(np.float16)
    # a weight that quantizes back to exactly these codes: w = (u - z) * s1
    w = (torch.from_numpy(u - zeros[:, None]).double() * torch.from_numpy(s1.astype(np.float64))[:, None]).float()
    lin = torch.nn.Linear(K, N, bias=False)
    lin.weight.data = w.clone()
    ql = cls.from_linear(lin, 4, -1, s1_scale=torch.from_numpy(s1).float(),
                         zeros=torch.from_numpy(zeros).to(torch.int8))
    np.savez_compressed(os.path.join(HERE, "w4a8_pack_per_chn.npz"),
                        codes=u.astype(np.uint8), zeros=zeros, s1=s1,
                        qweight=ql.qweight.numpy(), s1_scales=ql.s1_scales.numpy(),
                        s1_szeros=ql.s1_szeros.numpy())

    # ---- per-group g128: N=64, K=256 ------------------------------------------------
    N, K, G = 64, 256, 128
    u = rng.integers(0, 16, size=(N, K)).astype(np.int64)
    zg = rng.integers(0, 16, size=(N, K // G)).astype(np.int64)
    s2 = rng.integers(1, 9, size=(N, K // G)).astype(np.int64)
    s1 = rng.uniform(0.002, 0.02, size=(N,)).astype(np.float16)
    w8 = (u.reshape(N, K // G, G) - zg[:, :, None]) * s2[:, :, None]         # level-1 int8 weights
    w = (torch.from_numpy(w8.reshape(N, K)).double() * torch.from_numpy(s1.astype(np.float64))[:, None]).float()
    lin = torch.nn.Linear(K, N, bias=False)
    lin.weight.data = w.clone()
    ql = cls.from_linear(lin, 4, G, s1_scale=torch.from_numpy(s1).float(),
                         s2_scale=torch.from_numpy(s2), zeros=torch.from_numpy(zg))
    np.savez_compressed(os.path.join(HERE, "w4a8_pack_per_group.npz"),
                        codes=u.astype(np.uint8), zeros=zg, s2=s2, s1=s1,
                        qweight=ql.qweight.numpy(), s1_scales=ql.s1_scales.numpy(),
                        s2_scales=ql.s2_scales.numpy(), s2_zeros=ql.s2_zeros.numpy())
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
