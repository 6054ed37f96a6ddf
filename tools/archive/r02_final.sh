#!/bin/bash
# round 2, final GPU pass: full parity suite, smoke, bench line, per-kernel table of the decode step
set -u
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -12 > gpurun_out/final_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1
( time python bench.py ) > gpurun_out/final_bench.log 2>&1
tools/gpu_prof_cmd.sh final python $R/bench.py --steps 32 --warmup 4 --no-extras > gpurun_out/final_prof.log 2>&1
cat gpurun_out/final_pytest.log; tail -2 gpurun_out/final_smoke.log; grep -v amdgpu.ids gpurun_out/final_bench.log | cut -c1-6000; head -30 gpurun_out/prof_final_by_grid.md
