#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/r3c8.log
echo "== baseline lib (64-row tiles)" >> gpurun_out/r3c8.log
OMNI_SWEEP_OVERRIDES=0 python tools/mid_gemv_sweep.py >> gpurun_out/r3c8.log 2>&1
for l in lib_mid32_ar2_ring4 lib_mid32_ar2_ring8 lib_mid32_ar4_ring4; do
  echo "== $l" >> gpurun_out/r3c8.log
  OMNI_TUNE_LIB=tune_libs/$l.so OMNI_SWEEP_AR2=1 python tools/mid_gemv_sweep.py >> gpurun_out/r3c8.log 2>&1
done
python bench.py --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline (reverted carriers)', d['ms_per_step'], 'ms', d['value'], 'tok/s')" >> gpurun_out/r3c8.log 2>&1
