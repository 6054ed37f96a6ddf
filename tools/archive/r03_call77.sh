#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c77
(time python bench.py) > gpurun_out/r3c77/bench.log 2>&1
grep metric gpurun_out/r3c77/bench.log | cut -c1-200
