import sys, time, torch
sys.path.insert(0, '/root/repo')
from omniserve_amd.runtime import DecodeRunner, LlamaConfig
dev = torch.device("cuda:0")
cfg = LlamaConfig.llama2_70b(-1)
for comm in (None, "loopback", "loopback2"):
    r = DecodeRunner(cfg, 128, 1024, 40, dev, seed=3, tp_rank=0, tp_size=8, tp_comm=comm)
    for _ in range(4): r.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(16): r.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 16
    print(comm, "%.3f ms/step" % (dt * 1e3), flush=True)
    del r; torch.cuda.empty_cache()
