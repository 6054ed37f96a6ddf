#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c31
(timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -4) | tee gpurun_out/r3c31/tests.log
