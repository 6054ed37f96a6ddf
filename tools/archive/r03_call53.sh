#!/bin/bash
# tensor parallel: attention-side level-2 fusions (merge inside the quantiser, q / k / v from the qkv slabs) at level 1
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c53; O=gpurun_out/r3c53; rm -f $O/*.log
(timeout 900 python -m pytest tests/test_tp_gpu.py tests/test_runtime_gpu.py -x -q 2>&1 | tail -3) > $O/tests.log 2>&1
for v in 0 1 0 1; do
  echo "tp_l2_attn=$v $(OMNI_TP_L2_ATTN=$v timeout 300 python tools/tp_rank_steps.py 128 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/tests.log
done
cat $O/tests.log
