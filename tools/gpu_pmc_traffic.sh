#!/bin/bash
# HBM traffic per dispatch of the kernels matching a pattern, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (together they exceed the TCC counter slots and the collection aborts),
# both in KB; on gfx950 FETCH_SIZE reads half the bytes of a wide streaming read (double it).
#   tools/gpu_pmc_traffic.sh <kernel-name-substring> <command...>
# (inner timeouts are short: a hung collection must not eat the GPU budget)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out
PAT=$1; shift
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_traffic_$C
  (cd /tmp && timeout 90 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_traffic_$C -o t -- "$@") 2>&1 | grep -v "amdgpu.ids\|simple_timer\|output_stream\|tool.cpp" | tail -2
done
PAT=$PAT python - <<'PY'
import csv, glob, collections, os
pat = os.environ["PAT"]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/pmc_traffic_%s/**/*counter_collection.csv" % c, recursive=True):
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
rm -rf $O/pmc_traffic_FETCH_SIZE $O/pmc_traffic_WRITE_SIZE
