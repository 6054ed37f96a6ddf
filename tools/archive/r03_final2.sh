#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3final2; O=gpurun_out/r3final2
(time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $O/tests.log 2>&1
(timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > $O/smoke.log 2>&1
(time python bench.py) > $O/bench.log 2>&1
tail -n 6 $O/tests.log; cat $O/smoke.log; tail -n 4 $O/bench.log | cut -c1-300
