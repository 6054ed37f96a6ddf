"""In-graph per-kernel floor: chains of trivial dependent kernels (padding_offsets over empty sequences)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib
from tools.sweep_graph import graph_time_us
dev = torch.device("cuda:0")
lib = _lib.lib()
for nb in (1, 16, 512, 4096):
    cu = torch.zeros((nb + 1,), dtype=torch.int32, device=dev)
    out = torch.zeros((16,), dtype=torch.int32, device=dev)

    def fn(i):
        lib.omni_compute_padding_offsets(out.data_ptr(), cu.data_ptr(), nb, 1, 0, _lib.current_stream())

    print("chain of trivial kernels, %4d workgroups: %.2f us per kernel (graph)" % (nb, graph_time_us(fn, 200)), flush=True)
    # plain stream launches
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(50):
        fn(i)
    a.record()
    for i in range(2000):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    print("                          %4d workgroups: %.2f us per kernel (stream)" % (nb, a.elapsed_time(b) * 1e3 / 2000), flush=True)
