#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c28
timeout 300 python tools/qkv_split_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3c28/qkv_split.log
