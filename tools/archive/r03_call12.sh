#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/r3c12_tests.log 2>&1
(time python bench.py) > gpurun_out/r3c12_bench.log 2>&1
