"""configs[2] (Llama-3-8B g128, bs = 64) decode steps for rocprofv3: tools/gpu_prof_cmd.sh TAG python $PWD/tools/cfg2_steps.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd.runtime import DecodeRunner, LlamaConfig
dev = torch.device("cuda:0")
r = DecodeRunner(LlamaConfig.llama3_8b(128), 64, 1024, 60, dev, seed=0, fused=int(os.environ.get("OMNI_FUSED", "2")))
for _ in range(6): r.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(32): r.step()
torch.cuda.synchronize(); print("configs[2]: %.4f ms per step" % ((time.perf_counter() - t0) / 32 * 1e3))
