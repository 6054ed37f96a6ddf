#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c30; O=gpurun_out/r3c30
(timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -3) > $O/tests.log 2>&1
for mode in chn w8 grp; do
  timeout 300 python tools/gemm_ab.py $mode 2>&1 | grep -v amdgpu.ids > $O/ab_own_$mode.log
  OMNI_TUNE_LIB=tune_libs/libnoown.so timeout 300 python tools/gemm_ab.py $mode 2>&1 | grep -v amdgpu.ids > $O/ab_noown_$mode.log
done
OMNI_TUNE_LIB=tune_libs/libclk.so timeout 300 python tools/gemm_timeline.py 2>&1 | grep -v amdgpu.ids | head -8 > $O/timeline.log
cat $O/tests.log; for m in chn w8 grp; do echo "== $m own | noown"; paste -d'|' <(cut -c1-120 $O/ab_own_$m.log) <(cut -c58-140 $O/ab_noown_$m.log); done; cat $O/timeline.log | cut -c1-200
