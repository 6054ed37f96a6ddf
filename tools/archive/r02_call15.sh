#!/bin/bash
set -u
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_reference_layer_golden_gpu.py tests/test_runtime_gpu.py -m gpu -q --tb=short -s -k "golden" 2>&1 | grep -v amdgpu | tail -40
