#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c49; O=gpurun_out/r3c49; rm -f $O/*.log
(timeout 900 python -m pytest tests/test_rowfree_gpu.py -x -q -k "slabs" 2>&1 | tail -15) > $O/tests.log 2>&1
cat $O/tests.log
