#!/bin/bash
# round 3, third session, final pass: full GPU suite, smoke, bench line, rocprofv3 kernel-trace of the decode step + constants
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r3final4; O=gpurun_out/r3final4; R=$PWD
(time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/tests.log 2>&1
(timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > $O/smoke.log 2>&1
(time python bench.py) > $O/bench.log 2>&1
bash tools/r03_profile.sh r03_j > $O/profile.log 2>&1
tools/gpu_prof_cmd.sh r03_j_cfg2 python $R/bench.py --group-size 128 --batch 64 --steps 16 --warmup 4 --no-extras > $O/prof_cfg2.log 2>&1
cp gpurun_out/prof_r03_j_by_grid.md gpurun_out/prof_r03_j_cfg2_by_grid.md gpurun_out/r03_j_pmc.log gpurun_out/bench_constants.json $O/ 2>/dev/null
tail -n 4 $O/tests.log; cat $O/smoke.log; grep metric $O/bench.log | cut -c1-3500
