#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c58; O=gpurun_out/r3c58; rm -f $O/*.log
for mb in 0 12 0 12 24; do
  echo "tp rank bs128 prefetch_mb=$mb $(OMNI_PREFETCH_MB=$mb timeout 300 python tools/tp_rank_steps.py 128 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/steps.log
done
echo "default $(timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | head -2 | cut -c1-60 | tr '\n' ' ')" >> $O/steps.log
(timeout 600 python -m pytest tests/test_runtime_gpu.py -x -q 2>&1 | tail -2) >> $O/steps.log
cat $O/steps.log
