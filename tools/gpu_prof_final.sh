#!/bin/bash
# Round-end evidence: rocprofv3 --kernel-trace --stats of the DEFAULT bench command (so the roofline leg's launches are
# in the trace), then a per-(kernel, grid) table so that shapes sharing one kernel name can be read separately.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
R=$PWD
rm -rf $O/prof_final
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -o bench -- python $R/bench.py) 2>&1 | grep "^{\"metric\"" | tail -1 > $O/prof_final_bench_line.json
python - <<'PY'
import csv, collections
agg = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])
for r in csv.DictReader(open('gpurun_out/prof_final/bench_kernel_trace.csv')):
    n = r['Kernel_Name']
    if 'omni' not in n and 'Cijk' not in n and 'reduce_kernel' not in n:
        continue
    key = (n.replace('omni::', '')[:84], int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg[key]; a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
with open('gpurun_out/prof_final_by_grid.md', 'w') as f:
    f.write("| kernel | workgroups (x,y,z) | calls | avg us | min us | max us | total ms |\n|---|---|---|---|---|---|---|\n")
    for (n, gx, gy, gz), (c, t, mn, mx) in rows[:40]:
        f.write("| `%s` | %d,%d,%d | %d | %.2f | %.2f | %.2f | %.2f |\n" % (n, gx, gy, gz, c, t / c, mn, mx, t / 1e3))
print(open('gpurun_out/prof_final_by_grid.md').read()[:3500])
PY
cat $O/prof_final_bench_line.json | cut -c1-400
