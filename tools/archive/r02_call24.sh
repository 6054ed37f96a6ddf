#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/c24.log
for n in M32W4 M32W8; do
  echo "== $n" >> gpurun_out/c24.log
  OMNI_TUNE_LIB=tune_libs/lib_attn_$n.so timeout 300 python tools/attn_prefill_bench.py >> gpurun_out/c24.log 2>&1
done
timeout 900 python -m pytest tests/test_prefill_attn_gpu.py -m gpu -q --tb=short 2>&1 | tail -15 >> gpurun_out/c24.log
grep -v amdgpu.ids gpurun_out/c24.log
