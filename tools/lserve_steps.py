"""LServe decode driver (configs[3]) in isolation, for rocprofv3 --kernel-trace --stats: python tools/lserve_steps.py [kv4|kv8] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402

if os.environ.get("OMNI_TUNE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["OMNI_TUNE_LIB"])
from omniserve_amd.lserve_runtime import LServeDecodeRunner  # noqa: E402
from omniserve_amd.runtime import LlamaConfig  # noqa: E402

fmt = sys.argv[1] if len(sys.argv) > 1 else "kv8"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
r = LServeDecodeRunner(LlamaConfig.llama3_8b(-1), 1, 256000, steps + 16, dev, seed=7, kv_format=fmt)
for _ in range(8):
    r.step()
torch.cuda.synchronize()
per = {True: [], False: []}
for _ in range(steps):
    sel = r.steps_done == 0 or (r.context0 + r.steps_done + 1) % r.interval == 0     # the runner's refresh rule
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.step()
    torch.cuda.synchronize()
    per[sel].append(time.perf_counter() - t0)
print("%s: selector steps %.3f ms, other steps %.3f ms" % (fmt, 1e3 * sum(per[True]) / max(len(per[True]), 1),
                                                          1e3 * sum(per[False]) / max(len(per[False]), 1)))
