#!/bin/bash
# PMC passes over scripts/diag_wgrad.py (normal build): L2 hit rate, fabric bytes, L1 / TA busy of the weight-gradient kernels
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_wgrad
rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1
grep -o "TCP_[A-Z_0-9a-z]*\|TA_[A-Z_0-9a-z]*\|TCC_[A-Z_0-9a-z]*\|TD_[A-Z_0-9a-z]*" "$OUT/avail.txt" | sort -u > "$OUT/avail_mem.txt"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -- python3 scripts/diag_wgrad.py > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad" not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"].split("(")[0][-20:], r["Grid_Size"] if "Grid_Size" in r else "")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
