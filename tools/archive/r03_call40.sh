#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c40; O=gpurun_out/r3c40
(time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $O/tests.log 2>&1
(timeout 300 python tools/lserve_steps.py kv8 32 2>&1 | grep -v amdgpu.ids | tail -1) > $O/lserve.log 2>&1
(timeout 300 python tools/lserve_steps.py kv4 32 2>&1 | grep -v amdgpu.ids | tail -1) >> $O/lserve.log 2>&1
(time python bench.py) > $O/bench.log 2>&1
tail -n 6 $O/tests.log; cat $O/lserve.log; tail -n 3 $O/bench.log | cut -c1-1500
