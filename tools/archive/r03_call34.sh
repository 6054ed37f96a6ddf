#!/bin/bash
# W8A8 row-kernel-free forms (gemm_silu_w8a8 / gemm_partial_f16_w8a8 / sparse wide merge): parity, then LServe 256 K decode A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c34; O=gpurun_out/r3c34
(timeout 900 python -m pytest tests/test_rowfree_gpu.py tests/test_lserve_runtime_gpu.py -x -q 2>&1 | tail -5) > $O/tests.log 2>&1
for v in 1 0 1 0; do
  for fmt in kv8 kv4; do
    echo "rowfree=$v $(OMNI_LSERVE_ROWFREE=$v timeout 300 python tools/lserve_steps.py $fmt 32 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/steps.log
  done
done
cat $O/tests.log $O/steps.log
