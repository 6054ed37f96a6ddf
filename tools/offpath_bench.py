"""HBM rate of the off-path row kernels (csrc/offpath.hip) at a prefill-sized launch: 16384 tokens x 4096 (norms,
quantisers, dequantisers) and x 14336 (activations).  HIP events around 20 launches after 3 warm-ups, inputs rotated over
4 copies (> 256 MB MALL for the larger ones).  Prints one line per op: microseconds, algorithmic bytes, TB/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omniserve_backend.activation_ops as act
import omniserve_backend.fused_kernels as fk
import omniserve_backend.layernorm_ops as ln

dev = torch.device("cuda:0")
T, H, D = 16384, 4096, 14336
g = torch.Generator(device="cpu").manual_seed(0)


def timed(name, fn, nbytes, copies=4, iters=20):
    for i in range(3):
        fn(i % copies)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i % copies)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1000 / iters
    print("%-46s %9.1f us  %7.1f MB  %5.2f TB/s" % (name, us, nbytes / 1e6, nbytes / us / 1e6))


xs = [torch.randn((T, H), generator=g).half().to(dev) for _ in range(4)]
accs = [torch.randint(-60000, 60000, (T, H), generator=g, dtype=torch.int32).to(dev) for _ in range(4)]
res = [torch.randn((T, H), generator=g).half().to(dev) for _ in range(4)]
gus = [torch.randint(-40000, 40000, (T, 2 * D), generator=g, dtype=torch.int32).to(dev) for _ in range(2)]
xd = [torch.randn((T, D), generator=g).half().to(dev) for _ in range(4)]
w = (1 + 0.1 * torch.randn((H,), generator=g)).half().to(dev)
q8 = torch.empty((T, H), dtype=torch.int8, device=dev)
o16 = torch.empty((T, H), dtype=torch.float16, device=dev)
qd = torch.empty((T, D), dtype=torch.int8, device=dev)
od = torch.empty((T, D), dtype=torch.float16, device=dev)
tok = (0.0003 * torch.ones((T,))).half().to(dev)
scaling = torch.tensor([20.0], dtype=torch.float16, device=dev)
sc32 = torch.empty((T,), dtype=torch.float32, device=dev)
tmp = torch.empty((T, D), dtype=torch.float32, device=dev)

timed("invoke_quant (static)", lambda i: fk.invoke_quant(q8, xs[i], 0.04), T * H * 3)
timed("invoke_dequant", lambda i: fk.invoke_dequant(o16, accs[i], 0.003), T * H * 6)
timed("invoke_dequant_add_residual (per token)", lambda i: fk.invoke_dequant_add_residual(o16, accs[i], res[i], tok), T * H * 8)
timed("rms_norm (use_quant)", lambda i: ln.rms_norm(q8, xs[i], w, 1e-5, True), T * H * 3)
timed("rms_norm_general (per tensor)", lambda i: ln.rms_norm_general(q8, xs[i], w, scaling, 1e-5, False), T * H * 3)
timed("dequant_add_residual_rms_norm_quant", lambda i: ln.invoke_dequant_add_residual_rms_norm_quant(q8, accs[i], res[i], w, tok, 1e-6),
      T * H * 9)
timed("gelu_new", lambda i: act.gelu_new(od, xd[i]), T * D * 4)
timed("gelu_fast", lambda i: act.gelu_fast(od, xd[i]), T * D * 4)
timed("dequant_silu_and_mul_quant (static)", lambda i: act.invoke_dequant_silu_and_mul_quant(qd, gus[i], 1e-4, 1e-4, 0.05), T * D * 9,
      copies=2)
timed("dequant_silu_and_mul_quant (per token, + tmp)", lambda i: act.invoke_dequant_silu_and_mul_quant(qd, gus[i], 1e-4, 1e-4, sc32, tmp),
      T * D * 13, copies=2)
# the hot-path counterparts, for scale
s16 = torch.empty((T,), dtype=torch.float16, device=dev)
timed("[hot path] invoke_quant (per token)", lambda i: fk.invoke_quant(q8, xs[i], s16), T * H * 3)
timed("[hot path] rms_norm_general (per token)", lambda i: ln.rms_norm_general(q8, xs[i], w, s16, 1e-5, True), T * H * 3)
