#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c54; O=gpurun_out/r3c54; rm -f $O/*.log
(timeout 900 python -m pytest tests/test_tp_gpu.py -x -q 2>&1 | tail -5) > $O/tests.log 2>&1
cat $O/tests.log
