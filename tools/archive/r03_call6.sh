#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tools/r03_profile.sh r03_a
cp gpurun_out/bench_constants.json profiles/bench_constants.json 2>/dev/null
(time python bench.py) > gpurun_out/r3c6_bench_full.log 2>&1
