#!/bin/bash
# kernel-level durations of scripts/bench_wgrad_layers.py (rocprofv3 kernel trace): the copies and memsets are separate kernels there
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
rm -rf "$ROOT/gpurun_out/wgl" && mkdir -p "$ROOT/gpurun_out/wgl"
cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wgl -o wgl -- python3 scripts/bench_wgrad_layers.py > gpurun_out/wgl/out.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/wgl/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# sequence of wgrad kernels in launch order; group by consecutive identical (name, grid)
seq = [(r['Kernel_Name'].split('(')[0][-40:], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if 'wgrad' in r['Kernel_Name']]
groups = []
for n, gx, gy, gz, d in seq:
    key = (n, gx, gy, gz)
    if groups and groups[-1][0] == key: groups[-1][1].append(d)
    else: groups.append([key, [d]])
for key, ds in groups:
    ds = sorted(ds)
    print(key, 'n', len(ds), 'median us %.1f' % ds[len(ds) // 2])
PY
grep GFLOP gpurun_out/wgl/out.txt
