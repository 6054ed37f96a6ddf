#!/bin/bash
# mid-M (33..128) on the chunked prefill tile (64 x 256 per workgroup, shared LDS activation tile, split K): parity + sweep
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c36; O=gpurun_out/r3c36
(OMNI_MIDM_GEMM=1 timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "70b or mid_batches or bs64" 2>&1 | tail -3) > $O/tests.log 2>&1
OMNI_SWEEP_OVERRIDES=0 timeout 300 python tools/mid_gemv_sweep.py 2>&1 | grep -v amdgpu.ids > $O/sweep_default.log
OMNI_MIDM_GEMM=1 OMNI_SWEEP_OVERRIDES=0 timeout 300 python tools/mid_gemv_sweep.py 2>&1 | grep -v amdgpu.ids > $O/sweep_midm_auto.log
for sk in 1 2 4 8; do
OMNI_MIDM_GEMM=1 OMNI_MIDM_SK=$sk OMNI_SWEEP_OVERRIDES=0 timeout 300 python tools/mid_gemv_sweep.py 2>&1 | grep -v amdgpu.ids > $O/sweep_midm_sk$sk.log
done
cat $O/tests.log; for f in $O/sweep_*.log; do echo "== $f"; cat $f | cut -c1-100; done
