"""Dense causal prefill attention at 16 K x 32 heads, a few launches (for rocprofv3 --pmc)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flash_attn.flash_attn_interface import flash_attn_varlen_func
dev = torch.device("cuda:0"); L, Hq, Hk, D = 16384, 32, 8, 128
q = torch.randn((L, Hq, D), dtype=torch.float16, device=dev); k = torch.randn((L, Hk, D), dtype=torch.float16, device=dev); v = torch.randn_like(k)
cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
for _ in range(3):
    flash_attn_varlen_func(q, k, v, cu, cu, L, L, causal=True)
torch.cuda.synchronize(); print("done")
