"""Same-box A/B of decode steps with the mid-M GEMM kernel on (planner) / off:
python tools/step_midm_ab.py [group_size] [batch] [model: 8b|70b]   -> ms per step, alternating, three rounds"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402
from omniserve_amd.runtime import DecodeRunner, LlamaConfig  # noqa: E402

gs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
model = sys.argv[3] if len(sys.argv) > 3 else "8b"
dev = torch.device("cuda:0")
cfg = LlamaConfig.llama3_8b(gs) if model == "8b" else LlamaConfig.llama2_70b(gs)
lib = _lib.lib()


def build(mode):
    lib.omni_gemm_set_midm_override(mode, 0)        # plans are taken at graph capture
    r = DecodeRunner(cfg, bs, 1024, 200, dev, seed=0)
    for _ in range(6):
        r.step()
    torch.cuda.synchronize()
    return r


runners = {"midm planner": build(-1), "midm off": build(0)}
for rnd in range(3):
    for name, r in runners.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(32):
            r.step()
        torch.cuda.synchronize()
        print("round %d %-13s: %.4f ms per step" % (rnd, name, (time.perf_counter() - t0) / 32 * 1e3), flush=True)
