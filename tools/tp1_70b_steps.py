"""The whole Llama-2-70B (configs[4] at TP = 1) on one GPU at bs = 128, for rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd.runtime import DecodeRunner, LlamaConfig  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
r = DecodeRunner(LlamaConfig.llama2_70b(-1), bs, 1024, 24, dev, seed=3)
for _ in range(3):
    r.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(6):
    r.step()
torch.cuda.synchronize()
print("bs=%d: %.3f ms per step" % (bs, (time.perf_counter() - t0) / 6 * 1e3))
