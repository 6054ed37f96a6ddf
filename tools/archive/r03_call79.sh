#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c79; O=gpurun_out/r3c79; rm -f $O/*.log
for i in 1 2; do
echo "kv8 $(timeout 300 python tools/lserve_steps.py kv8 32 2>&1 | grep -v amdgpu.ids | tail -1) | kv4 $(timeout 300 python tools/lserve_steps.py kv4 32 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/steps.log
done
python - >> $O/steps.log 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
import bench
r = bench.lserve_leg(torch.device('cuda:0'))
print({k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if 'ms' in a or 'stage_s' in a}) for k, v in r.items() if k in ('kv4', 'kv8')})
PY
cat $O/steps.log | grep -v amdgpu
