#!/bin/bash
set -u
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
tools/gpu_pmc_generic.sh prefill_attn python $R/tools/attn16k.py > gpurun_out/c19.log 2>&1
grep -v "amdgpu.ids\|Opened result" gpurun_out/c19.log | tail -30
