#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c35; O=gpurun_out/r3c35
(timeout 900 python -m pytest tests/test_rowfree_gpu.py tests/test_lserve_runtime_gpu.py tests/test_reference_lserve_layer_golden_gpu.py -q 2>&1 | tail -8) > $O/tests.log 2>&1
cat $O/tests.log
