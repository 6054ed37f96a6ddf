#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/r3c9.log
(timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_runtime_gpu.py tests/test_tp_gpu.py tests/test_elementwise_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r3c9_tests.log 2>&1
echo "== narrow on (default)" >> gpurun_out/r3c9.log
OMNI_SWEEP_OVERRIDES=0 python tools/mid_gemv_sweep.py >> gpurun_out/r3c9.log 2>&1
echo "== narrow off" >> gpurun_out/r3c9.log
OMNI_GEMV_NARROW=0 OMNI_SWEEP_OVERRIDES=0 python tools/mid_gemv_sweep.py >> gpurun_out/r3c9.log 2>&1
python - >> gpurun_out/r3c9.log 2>&1 <<'PY'
import sys, torch, argparse
sys.path.insert(0, '.')
import bench
args = argparse.Namespace(context=1024, batch=16, no_fused=False, fused_level=3)
dev = torch.device('cuda:0')
print('configs2', bench.configs2_leg(args, dev))
print('tp rank', bench.tp_rank_leg(dev))
PY
