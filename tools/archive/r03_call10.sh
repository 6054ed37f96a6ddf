#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/r3c10.log
for nv in 1 0 1 0; do
OMNI_GEMV_NARROW=$nv python - >> gpurun_out/r3c10.log 2>&1 <<'PY'
import sys, os, time, torch
sys.path.insert(0, '.')
from omniserve_amd.runtime import DecodeRunner, LlamaConfig
dev = torch.device('cuda:0')
def run(cfg, B, steps=32, warm=6, **kw):
    r = DecodeRunner(cfg, B, 1024, steps + warm + 4, dev, seed=77, **kw)
    for _ in range(warm): r.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): r.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    del r; torch.cuda.empty_cache()
    return dt * 1e3
print('narrow', os.environ['OMNI_GEMV_NARROW'], 'configs2 g128 bs64: %.3f ms' % run(LlamaConfig.llama3_8b(128), 64),
      ' 70B tp8 rank bs128: %.3f ms' % run(LlamaConfig.llama2_70b(-1), 128, steps=16, warm=4, fused=1, tp_rank=0, tp_size=8),
      ' per-chn bs64: %.3f ms' % run(LlamaConfig.llama3_8b(-1), 64), ' bs32: %.3f ms' % run(LlamaConfig.llama3_8b(-1), 32))
PY
done
