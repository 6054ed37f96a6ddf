#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c51; O=gpurun_out/r3c51; rm -f $O/*.log
(timeout 900 python -m pytest tests/test_rowfree_gpu.py -x -q -k "slabs" 2>&1 | tail -3) > $O/tests.log 2>&1
for v in 0 1 0 1; do
  echo "qkv_slabs=$v $(OMNI_QKV_SLABS=$v timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | head -2 | cut -c1-60 | tr '\n' ' ') | $(OMNI_QKV_SLABS=$v timeout 300 python tools/lserve_steps.py kv8 32 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/steps.log
done
cat $O/tests.log $O/steps.log
