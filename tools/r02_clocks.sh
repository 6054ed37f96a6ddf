#!/bin/bash
# sclk / power under the three loads (rocm-smi sampled while the loop runs)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/clocks.log
for what in attn gemm decode; do
  echo "== $what" >> gpurun_out/clocks.log
  python tools/attn_loop.py $what 7 >> gpurun_out/clocks.log 2>&1 &
  PID=$!
  sleep 4
  for i in 1 2 3; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|mclk" | head -4 >> gpurun_out/clocks.log
    sleep 0.7
  done
  wait $PID
done
grep -v amdgpu.ids gpurun_out/clocks.log
