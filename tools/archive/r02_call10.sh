#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/c10.log
OMNI_SWEEP_WIDE=1 timeout 600 python tools/mid_gemv_sweep.py >> gpurun_out/c10.log 2>&1
grep -v amdgpu.ids gpurun_out/c10.log
