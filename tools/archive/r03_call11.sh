#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/r3c11.log
for v in 1 0 1 0; do
OMNI_W8_SMALL_SPLIT=$v python - >> gpurun_out/r3c11.log 2>&1 <<'PY'
import sys, os, time, torch
sys.path.insert(0, '.')
from omniserve_amd.lserve_runtime import LServeDecodeRunner
from omniserve_amd.runtime import LlamaConfig
dev = torch.device('cuda:0')
out = []
for fmt in ('kv8',):
    r = LServeDecodeRunner(LlamaConfig.llama3_8b(-1), 1, 256000, 48, dev, seed=7, kv_format=fmt)
    for _ in range(8): r.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(32): r.step()
    torch.cuda.synchronize(); out.append('%s %.3f ms/step' % (fmt, (time.perf_counter() - t0) / 32 * 1e3))
    del r; torch.cuda.empty_cache()
print('w8_small_split', os.environ['OMNI_W8_SMALL_SPLIT'], out)
PY
done
(timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_lserve_runtime_gpu.py -x -q 2>&1 | tail -3) >> gpurun_out/r3c11.log 2>&1
