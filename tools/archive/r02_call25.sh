#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_runtime_gpu.py tests/test_elementwise_gpu.py tests/test_ckpt_gpu.py tests/test_reference_layer_golden_gpu.py -m gpu -q --tb=short -x 2>&1 | grep -v amdgpu | tail -12
timeout 300 python bench.py --group-size 128 --batch 64 --steps 32 --warmup 4 --no-extras 2>&1 | grep -v amdgpu | cut -c1-400
OMNI_FUSED_LEVEL=1 timeout 300 python bench.py --group-size 128 --batch 64 --steps 32 --warmup 4 --no-extras --fused-level 1 2>&1 | grep -v amdgpu | cut -c1-400
