#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c27
(timeout 900 python -m pytest tests/test_row_dtypes_gpu.py tests/test_elementwise_gpu.py -x -q 2>&1 | tail -12) | tee gpurun_out/r3c27/tests.log
