#!/bin/bash
# (round 1: the first run of this script hung inside rocprofv3 until the limit -- keep the inner timeout short)
# HBM traffic (FETCH_SIZE / WRITE_SIZE, KB) per dispatch for the kernels matching a pattern: tools/gpu_pmc_traffic.sh <pattern> <command...>
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out
PAT=$1; shift
rm -rf $O/pmc_traffic
(cd /tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_traffic -o t -- "$@") 2>&1 | grep -v "amdgpu.ids\|simple_timer\|output_stream" | tail -2
PAT=$PAT python - <<'PY'
import csv, glob, collections, os
pat = os.environ["PAT"]
for f in glob.glob("gpurun_out/pmc_traffic/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        if pat not in row.get("Kernel_Name", ""):
            continue
        k = (row.get("Kernel_Name")[:60], row.get("Grid_Size"), row.get("Counter_Name"))
        agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0))
    for k in sorted(agg):
        n, v = agg[k]
        print("%-62s grid %-9s %-12s dispatches %3d mean %14.1f KB" % (k[0], k[1], k[2], n, v / n))
PY
