#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary command; prints the top kernels: tools/gpu_stats_cmd.sh <tag> <command...>
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out
TAG=$1; shift
rm -rf $O/stats_$TAG
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$TAG -o s -- "$@") 2>&1 | grep -v "amdgpu.ids\|simple_timer\|output_stream" | tail -3
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/stats_$TAG/s_kernel_stats.csv")))
print("%-100s %8s %10s %10s" % ("kernel", "calls", "avg us", "total ms"))
for r in rows[:22]:
    print("%-100s %8s %10.2f %10.2f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
