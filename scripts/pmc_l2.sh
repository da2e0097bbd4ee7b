#!/bin/bash
# L2 hit rate of the conv kernels (separate --pmc pass, kernel-trace only)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_l2
rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/l2" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "l2", "**", "*counter_collection.csv"), recursive=True)[0]
acc = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].replace("dgp::", "")[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:6]:
    h, m = d.get("TCC_HIT_sum", 0), d.get("TCC_MISS_sum", 0)
    print("%-62s L2 hit %.3f  (hits %.3g misses %.3g)" % (k, h / max(h + m, 1), h, m))
PY
