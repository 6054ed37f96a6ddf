#!/bin/bash
set -u
cd "$(dirname "$0")/.."
for kv in 0 1 0 1; do
  echo "== OMNI_PREFETCH_KV=$kv"
  OMNI_PREFETCH_KV=$kv timeout 300 python bench.py --no-extras --steps 128 --warmup 8 2>&1 | grep -v amdgpu | grep -o '"value": [0-9.]*, "unit": "tokens/s", "n_gpus": 1, "steps": 128, "warmup": 8, "ms_per_step": [0-9.]*'
done
timeout 600 python -m pytest tests/test_runtime_gpu.py tests/test_reference_layer_golden_gpu.py -m gpu -q --tb=short 2>&1 | grep -v amdgpu | tail -4
