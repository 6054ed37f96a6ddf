#!/bin/bash
# attention of layer l touches layer l + 1's page-table window: A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c67; O=gpurun_out/r3c67; rm -f $O/*.log
for v in 0 1 0 1; do
  echo "hint_tables=$v $(OMNI_HINT_TABLES=$v timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | head -2 | cut -c1-60 | tr '\n' ' ')" >> $O/steps.log
done
(timeout 600 python -m pytest tests/test_runtime_gpu.py tests/test_kv4_gpu.py tests/test_rowfree_gpu.py tests/test_reference_layer_golden_gpu.py -x -q 2>&1 | tail -2) >> $O/steps.log
cat $O/steps.log
