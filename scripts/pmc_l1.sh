#!/bin/bash
# L1 (TCP) vs L2 (TCC) requests of the conv kernels on one stream, per tier: how many of the L2 -> LDS lines are shared between the two
# co-resident workgroups of a CU (separate --pmc passes, kernel-trace only).  Usage: scripts/pmc_l1.sh [f16|parity]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; T=${1:-f16}
OUT=$ROOT/gpurun_out/pmc_l1_$T
rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
rocprofv3 -L 2>/dev/null | grep -oE "TCP_[A-Z_0-9]+(_sum)?" | sort -u | tr '\n' ' ' > "$OUT/tcp_counters.txt"
for pass in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/$tag" -- python3 scripts/bench_tier.py $T --steps 3 > "$OUT/$tag.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("dgp::", "")[:78]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("TCC_REQ_sum", 0))[:8]:
    per = {c: v / max(n[(k, c)], 1) for c, v in d.items()}
    print("%-80s per launch: TCP accesses %.3g, TCP->TCC reads %.3g (%.2f of accesses), TCC req %.3g hit %.3f" % (
        k, per.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0), per.get("TCP_TCC_READ_REQ_sum", 0),
        per.get("TCP_TCC_READ_REQ_sum", 0) / max(per.get("TCP_TOTAL_CACHE_ACCESSES_sum", 1), 1), per.get("TCC_REQ_sum", 0),
        per.get("TCC_HIT_sum", 0) / max(per.get("TCC_HIT_sum", 0) + per.get("TCC_MISS_sum", 0), 1)))
PY
cat "$OUT/tcp_counters.txt" | cut -c1-600
