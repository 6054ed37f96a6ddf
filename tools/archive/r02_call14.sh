#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_tp_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -30
