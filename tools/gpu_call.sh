#!/bin/bash
# One parametrised runner for a gpurun call (replaces the per-call scripts of rounds 1-3).  Usage, from the repo root on the GPU box:
#     tools/gpu_call.sh TAG RECIPE [RECIPE ...]
# Recipes (recipe number i writes gpurun_out/TAG/<i>_<recipe>.log and prints its tail):
#     suite            python -m pytest tests -m gpu -x -q
#     suite:EXPR       ... -k EXPR
#     smoke            __graft_entry__.smoke()
#     bench            python bench.py                      (the driver's default line)
#     bench:ARGS       python bench.py ARGS                 (e.g. "bench:--no-extras --steps 32")
#     profile          rocprofv3 --kernel-trace of the decode step -> per-(kernel, grid) table, PMC traffic of the dominant
#                      kernel, profiles/bench_constants.json inputs (tools/gpu_profile.sh TAG)
#     py:SCRIPT ARGS   python SCRIPT ARGS                   (a tools/*.py probe)
#     sh:CMD           bash -c CMD
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
i=0
for r in "$@"; do
  i=$((i + 1))
  name=${r%%:*}; arg=""; [[ "$r" == *:* ]] && arg=${r#*:}
  log=$O/${i}_$(echo "$name" | tr -c 'A-Za-z0-9_\n' '_').log
  case $name in
    suite)   if [ -n "$arg" ]; then (time timeout 1500 python -m pytest tests -m gpu -x -q -k "$arg" 2>&1 | tail -15) > $log 2>&1
             else (time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $log 2>&1; fi ;;
    smoke)   (timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5) > $log 2>&1 ;;
    bench)   (time timeout 1200 python bench.py $arg) > $log 2>&1 ;;
    profile) bash tools/gpu_profile.sh $TAG > $log 2>&1; cp gpurun_out/prof_${TAG}_by_grid.md gpurun_out/${TAG}_pmc.log gpurun_out/bench_constants.json $O/ 2>/dev/null ;;
    py)      (time timeout 1200 python $arg) > $log 2>&1 ;;
    sh)      (time timeout 1200 bash -c "$arg") > $log 2>&1 ;;
    *)       echo "unknown recipe $r" ;;
  esac
  echo "== $r"; tail -n 12 $log | cut -c1-4000
done
