#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_rowfree_gpu.py tests/test_runtime_gpu.py tests/test_reference_layer_golden_gpu.py tests/test_kv4_gpu.py -x -q 2>&1 | tail -15) > gpurun_out/r3c3_tests.log 2>&1
tools/gpu_prof_cmd.sh r3c3_l3 python $R/bench.py --steps 32 --warmup 4 --no-extras --fused-level 3 > gpurun_out/r3c3_prof_l3.log 2>&1
tools/gpu_prof_cmd.sh r3c3_l2 python $R/bench.py --steps 32 --warmup 4 --no-extras --fused-level 2 > gpurun_out/r3c3_prof_l2.log 2>&1
