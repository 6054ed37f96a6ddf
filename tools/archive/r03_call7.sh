#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/r3c7_ab.log
(OMNI_PF3="0.2,0.2,0.2,0.2,0.3,64" timeout 900 python -m pytest tests/test_runtime_gpu.py tests/test_rowfree_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r3c7_tests.log 2>&1
for v in "0,0,0,0,0,64" "8.4,0,0,40,0,64" "8.4,8,12,20,0,64" "8.4,8,12,20,30,64" "8.4,0,0,40,30,64" "8.4,8,12,20,30,128" "8.4,12,16,24,30,64" "8.4,8,12,20,16,64"; do
  OMNI_PF3=$v python bench.py --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PF3=$v', d['ms_per_step'], 'ms', d['value'], 'tok/s')" >> gpurun_out/r3c7_ab.log 2>&1
done
