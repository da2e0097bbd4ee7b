#!/bin/bash
# rocprofv3 kernel stats of the training-step benchmark on the 16-bit tier (scripts/bench_train.py <steps> <warm-up> f16)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_train_f16
rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
# per-kernel durations with the weight gradients on the chain's stream (DGP_WGRAD_OVERLAP=0): kernels that share the chip with another
# stream's kernels would each be charged the shared time
export DGP_WGRAD_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 scripts/bench_train.py 12 2 f16 > "$OUT/bench.log" 2>&1
unset DGP_WGRAD_OVERLAP
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 14.0       # 2 warm-up (the first one on the parity path) + 12 timed
for r in rows[:26]:
    print("%-100s calls %5s  %8.3f ms/step  avg %9.1f us  %5s%%" % (r["Name"].replace("dgp::", "").replace("(anonymous namespace)::", "")[:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
grep "^{" "$OUT/bench.log" | cut -c1-200
echo "# the same step without the profiler, 60 steps: 16-bit tier, parity tier, alternating"
for t in f16 parity f16 parity; do python3 scripts/bench_train.py 60 8 $t | cut -c1-140; done
