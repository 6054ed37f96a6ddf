#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c70
timeout 900 python tools/split_sweep_cfgs.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3c70/sweep.log
