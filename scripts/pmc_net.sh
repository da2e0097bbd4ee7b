#!/bin/bash
# SQ counters of the conv kernels over a bench.py run (separate --pmc passes, kernel-trace only).  Usage: pmc_net.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_net
rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/log$i.txt" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(sys.argv[1], "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("dgp::", "")
        if "conv_igemm_split_ls" in k:
            a = acc[k[:70]][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c, (v, n) in sorted(acc[k].items()):
        print("    %-28s %.6g per dispatch (%d)" % (c, v / n, n))
PY
