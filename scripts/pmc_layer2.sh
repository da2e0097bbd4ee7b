#!/bin/bash
# SQ counters of one conv layer shape under one tile (separate --pmc passes, kernel-trace only).  Usage: pmc_layer2.sh <layer filter> <tile:label>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_layer2
rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
export SWEEP_TILES=${2:-16:H3k32}
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -- python3 scripts/split_sweep.py "$1" > "$OUT/log$i.txt" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(sys.argv[1], "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_igemm_split_ls" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print("%-28s %.6g per dispatch (%d dispatches)" % (k, v / n, n))
PY
