#!/bin/bash
# dense decode attention: a GQA group of 8 in one workgroup (G = 8).  Parity, then the 70B TP=8 rank step A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c69; O=gpurun_out/r3c69; rm -f $O/*.log
(timeout 900 python -m pytest tests/test_kv4_gpu.py tests/test_edge_cases_gpu.py tests/test_rowfree_gpu.py tests/test_tp_gpu.py tests/test_runtime_gpu.py -x -q 2>&1 | tail -3) > $O/tests.log 2>&1
for v in 4 8 4 8; do
  echo "max_g=$v $(OMNI_DECODE_MAX_G=$v timeout 300 python tools/tp_rank_steps.py 128 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/tests.log
done
cat $O/tests.log
