#!/bin/bash
# round 3, call 13: prefill GEMM epilogue rework -- parity of the main build, A/B of head / E1 (main) / E2 (exact kernel)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; O=gpurun_out/r3c13; mkdir -p $O
(timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -5) > $O/tests_gemm.log 2>&1
for lib in head e2; do
  for mode in chn; do
    OMNI_TUNE_LIB=tune_libs/lib$lib.so timeout 300 python tools/gemm_ab.py $mode > $O/ab_${lib}_$mode.log 2>&1
  done
done
timeout 300 python tools/gemm_ab.py chn --int-mm > $O/ab_main_chn.log 2>&1
for mode in grp w8; do
  OMNI_TUNE_LIB=tune_libs/libhead.so timeout 300 python tools/gemm_ab.py $mode > $O/ab_head_$mode.log 2>&1
  timeout 300 python tools/gemm_ab.py $mode > $O/ab_main_$mode.log 2>&1
done
tail -n 8 $O/*.log
