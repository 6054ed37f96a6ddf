#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary command, reduced to a per-(kernel, grid) table:
#   tools/gpu_prof_cmd.sh <tag> <command ...>      -> gpurun_out/prof_<tag>/ (raw), gpurun_out/prof_<tag>_by_grid.md
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=$1; shift
mkdir -p gpurun_out
O=$PWD/gpurun_out
R=$PWD
rm -rf $O/prof_$TAG
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o run -- "$@") > $O/prof_${TAG}_stdout.log 2>&1
grep -v amdgpu.ids $O/prof_${TAG}_stdout.log | tail -3 | cut -c1-600
TAG=$TAG python - <<'PY'
import csv, collections, os
tag = os.environ["TAG"]
agg = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])
for r in csv.DictReader(open('gpurun_out/prof_%s/run_kernel_trace.csv' % tag)):
    n = r['Kernel_Name']
    key = (n.replace('omni::', '')[:100], int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), int(r['Grid_Size_Y']) // max(int(r['Workgroup_Size_Y']), 1), int(r['Grid_Size_Z']) // max(int(r['Workgroup_Size_Z']), 1))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg[key]; a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
with open('gpurun_out/prof_%s_by_grid.md' % tag, 'w') as f:
    f.write("| kernel | workgroups (x,y,z) | calls | avg us | min us | max us | total ms |\n|---|---|---|---|---|---|---|\n")
    for (n, gx, gy, gz), (c, t, mn, mx) in rows[:48]:
        f.write("| `%s` | %d,%d,%d | %d | %.2f | %.2f | %.2f | %.2f |\n" % (n, gx, gy, gz, c, t / c, mn, mx, t / 1e3))
print(open('gpurun_out/prof_%s_by_grid.md' % tag).read()[:6000])
PY
# keep the merge-back small: the raw trace can be tens of MB
rm -f $O/prof_$TAG/*_kernel_trace.csv $O/prof_$TAG/*agent_info.csv 2>/dev/null
