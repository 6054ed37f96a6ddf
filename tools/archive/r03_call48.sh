#!/bin/bash
# 32-/64-row GEMV tiles: B-operand reads of step s + 1 in front of the MFMAs of step s.  Parity, sweep, steps.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c48; O=gpurun_out/r3c48; rm -f $O/*.log
(timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -3) > $O/tests.log 2>&1
for lib in pipeb0 new pipeb0 new; do
  if [ $lib = new ]; then unset OMNI_TUNE_LIB; else export OMNI_TUNE_LIB=tune_libs/lib$lib.so; fi
  echo "== $lib" >> $O/sweep.log
  OMNI_SWEEP_OVERRIDES=0 timeout 300 python tools/mid_gemv_sweep.py 2>&1 | grep -v amdgpu.ids | cut -c1-70 >> $O/sweep.log
  echo "$lib $(timeout 300 python tools/tp_rank_steps.py 128 2>&1 | grep -v amdgpu.ids | tail -1) | $(timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | sed -n 2p | cut -c1-60)" >> $O/steps.log
done
cat $O/tests.log $O/sweep.log $O/steps.log
