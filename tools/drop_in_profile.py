"""Host profile (cProfile) of the eager reference-call-sequence decode step (bench.py `drop_in`): where the Python time goes."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd.runtime import DecodeRunner, LlamaConfig  # noqa: E402

dev = torch.device("cuda:0")
r = DecodeRunner(LlamaConfig.llama3_8b(-1), 16, 1024, 80, dev, seed=99, use_graph=False, fused=0)
for _ in range(4):
    r.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(24):
    r.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
