"""The protocol's prefill stage (one prompt of 1024 tokens per sequence through DecodeRunner.prefill) in isolation, for
rocprofv3 --kernel-trace: python tools/prefill_steps.py [batch=64] [group_size=-1] [reps=2]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd.runtime import DecodeRunner, LlamaConfig  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
gs = int(sys.argv[2]) if len(sys.argv) > 2 else -1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
r = DecodeRunner(LlamaConfig.llama3_8b(gs), bs, 1024, 16, dev, seed=4321, fused=1)
r.prefill(1024)
torch.cuda.synchronize()
for _ in range(reps):
    t0 = time.perf_counter()
    r.prefill(1024)
    torch.cuda.synchronize()
    print("bs=%d: prefill %.2f ms = %.0f tokens/s" % (bs, (time.perf_counter() - t0) * 1e3, bs * 1024 / (time.perf_counter() - t0)))
