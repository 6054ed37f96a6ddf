#!/bin/bash
# round 3, final pass: full GPU suite, smoke, bench line, rocprofv3 kernel-trace summaries (decode step + prefill GEMMs)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r3final; O=gpurun_out/r3final; R=$PWD
(time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/tests.log 2>&1
(timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > $O/smoke.log 2>&1
(time python bench.py) > $O/bench.log 2>&1
tools/gpu_prof_cmd.sh r03_f_step python $R/bench.py --steps 32 --warmup 4 --no-extras > $O/prof_step.log 2>&1
tools/gpu_prof_cmd.sh r03_f_gemm python $R/tools/gemm_ab.py chn > $O/prof_gemm.log 2>&1
cp gpurun_out/prof_r03_f_step_by_grid.md gpurun_out/prof_r03_f_gemm_by_grid.md $O/ 2>/dev/null
tail -n 4 $O/tests.log; cat $O/smoke.log; tail -n 4 $O/bench.log | cut -c1-400; head -16 $O/prof_r03_f_gemm_by_grid.md
