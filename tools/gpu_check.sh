#!/bin/bash
# One-shot GPU validation run (invoked through gpurun): layout probe, parity tests, smoke, bench,
# GEMV plan sweep, rocprofv3 kernel stats.  Everything lands under gpurun_out/.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
echo "== probe" ; (hipcc --offload-arch=gfx950 -w tools/mfma_probe.hip -o /tmp/probe && timeout 60 /tmp/probe) 2>&1 | tee $O/probe.log
echo "== pytest" ; timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider 2>&1 | tail -80 | tee $O/pytest.log
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -20 | tee $O/smoke.log
echo "== bench" ; timeout 600 python bench.py --steps 64 --warmup 8 2>&1 | tail -20 | tee $O/bench.log
echo "== sweep" ; timeout 900 python tools/sweep_graph.py 16 2>&1 | tail -120 | tee $O/sweep_graph.log
echo "== debug" ; timeout 300 python tools/debug_norm.py 2>&1 | tail -40 | tee $O/debug_norm.log
echo "== rocprof" ; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof -o bench -- python $OLDPWD/bench.py --steps 32 --warmup 4 --no-extras) 2>&1 | tail -15 | tee $O/rocprof.log
ls -R $O/prof 2>/dev/null | head -30
echo "== pmc" ; (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OLDPWD/$O/pmc_fetch -o gemv -- python $OLDPWD/tools/gemv_loop.py) 2>&1 | tail -3 | tee $O/pmc.log
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OLDPWD/$O/pmc_write -o gemv -- python $OLDPWD/tools/gemv_loop.py) 2>&1 | tail -3 | tee -a $O/pmc.log
ls $O/pmc_fetch $O/pmc_write 2>/dev/null
