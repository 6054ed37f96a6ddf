#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c26
timeout 300 python tools/w8_decode_sweep.py 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3c26/w8_sweep.log
