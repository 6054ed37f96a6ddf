#!/bin/bash
# rocprofv3 kernel stats of the bench decode loop, compact table to stdout; raw CSV under gpurun_out/prof.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
R=$PWD
rm -rf $O/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 32 --warmup 4 --no-extras "$@") 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof/bench_kernel_stats.csv')))
for r in rows[:14]:
    n = r['Name'].replace('omni::', '').replace('_ZN4omni', '')
    print("%-72s calls %6s avg %9.2f us  %6s%%" % (n[:72], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
