#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c21; O=gpurun_out/r3c21
(time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/tests.log 2>&1
(time python bench.py) > $O/bench.log 2>&1
for lib in m64b3; do OMNI_TUNE_LIB=tune_libs/lib$lib.so timeout 300 python tools/gemm_ab.py chn 2>&1 | grep -v amdgpu.ids | head -4 > $O/ab_$lib.log; done
tail -n 12 $O/tests.log; tail -n 5 $O/bench.log | cut -c1-3000; cat $O/ab_m64b3.log
