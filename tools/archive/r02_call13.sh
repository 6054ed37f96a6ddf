#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/c13.log
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_elementwise_gpu.py tests/test_runtime_gpu.py tests/test_tp_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -8 >> gpurun_out/c13.log
OMNI_SWEEP_OVERRIDES=0 timeout 300 python tools/mid_gemv_sweep.py >> gpurun_out/c13.log 2>&1
timeout 300 python tools/tp_rank_steps.py 128 >> gpurun_out/c13.log 2>&1
timeout 300 python bench.py --group-size 128 --batch 64 --steps 32 --warmup 4 --no-extras >> gpurun_out/c13.log 2>&1
grep -v amdgpu.ids gpurun_out/c13.log | cut -c1-600
