#!/bin/bash
set -u
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_persistent_gpu.py tests/test_prefill_attn_gpu.py -m gpu -q --tb=short 2>&1 | grep -v amdgpu | tail -30
