#!/bin/bash
# o_proj's idle rider workgroups prefetch the head of down_proj's weights: A/B over the budget
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c55; O=gpurun_out/r3c55; rm -f $O/*.log
for mb in 0 4 8 16 0 8; do
  echo "rider_pf_mb=$mb $(OMNI_RIDER_PF_MB=$mb timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | head -1 | cut -c1-60)" >> $O/steps.log
done
(OMNI_RIDER_PF_MB=8 timeout 600 python -m pytest tests/test_runtime_gpu.py tests/test_rowfree_gpu.py -x -q 2>&1 | tail -2) >> $O/steps.log
cat $O/steps.log
