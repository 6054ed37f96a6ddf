#!/bin/bash
# round 3, third session: very last pass (suite, smoke, bench) on the committed defaults
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3final6; O=gpurun_out/r3final6
(time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/tests.log 2>&1
(timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > $O/smoke.log 2>&1
(time python bench.py) > $O/bench.log 2>&1
grep "passed\|failed" $O/tests.log; tail -1 $O/smoke.log; grep metric $O/bench.log | cut -c1-300
