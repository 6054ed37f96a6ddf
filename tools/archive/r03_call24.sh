#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c24
timeout 300 python tools/gemv_balance.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3c24/balance.log
