"""One TP=8 rank of Llama-2-70B (configs[4]) on one GPU, collectives skipped, for rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd.runtime import DecodeRunner, LlamaConfig  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
if os.environ.get("OMNI_MIDM"):      # planner override of the mid-M kernel (-1 heuristic, 0 never, 1 wherever legal); plans are taken at capture
    from omniserve_amd import _lib
    _lib.lib().omni_gemm_set_midm_override(int(os.environ["OMNI_MIDM"]), 0)
r = DecodeRunner(LlamaConfig.llama2_70b(-1), bs, 1024, 40, dev, seed=3, fused=int(os.environ.get("OMNI_FUSED", "3")), tp_rank=0, tp_size=8)
for _ in range(4):
    r.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    r.step()
torch.cuda.synchronize()
print("bs=%d: %.3f ms per step" % (bs, (time.perf_counter() - t0) / 8 * 1e3))
