#!/bin/bash
# rocprofv3 kernel stats of the training-step benchmark
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_train
rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
# per-kernel durations are taken with the weight gradients on the chain's stream (DGP_WGRAD_OVERLAP=0): kernels that share the chip
# with another stream's kernels would each be charged the shared time
export DGP_WGRAD_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 scripts/bench_train.py 12 > "$OUT/bench.log" 2>&1
unset DGP_WGRAD_OVERLAP
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 14.0
for r in rows[:22]:
    print("%-80s calls %5s  %8.3f ms/step  avg %9.1f us  %5s%%" % (r["Name"].replace("dgp::", "")[:80], r["Calls"], float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
grep "^{" "$OUT/bench.log" | cut -c1-200
echo "# the same step without the profiler, weight gradients on the second stream (default) and on the chain's stream:"
python3 scripts/bench_train.py 10 | cut -c1-400
DGP_WGRAD_OVERLAP=0 python3 scripts/bench_train.py 10 | cut -c1-120
