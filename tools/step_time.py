"""ms per decode step of one configuration, for same-box A/Bs of library variants (honours OMNI_TUNE_LIB):
python tools/step_time.py [group_size=-1] [batch=16] [rounds=4]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402

if os.environ.get("OMNI_TUNE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["OMNI_TUNE_LIB"])
from omniserve_amd.runtime import DecodeRunner, LlamaConfig  # noqa: E402

gs = int(sys.argv[1]) if len(sys.argv) > 1 else -1
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
r = DecodeRunner(LlamaConfig.llama3_8b(gs), bs, 1024, 64 * rounds + 16, dev, seed=0)
for _ in range(8):
    r.step()
torch.cuda.synchronize()
ts = []
for _ in range(rounds):
    t0 = time.perf_counter()
    for _ in range(64):
        r.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 64 * 1e3)
print("%s g%d bs%d: %s ms per step" % (os.path.basename(os.environ.get("OMNI_TUNE_LIB", "shipped")), gs, bs, " ".join("%.4f" % t for t in ts)))
