#!/bin/bash
# PMC diagnosis of one conv layer shape: scripts/pmc_layer.sh <tag> <layer-substring>
set -u
TAG=${1:-x}; LAYER=${2:-b4.conv2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/a" -- python3 scripts/conv_sweep.py "$LAYER" > "$OUT/a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d "$OUT/b" -- python3 scripts/conv_sweep.py "$LAYER" > "$OUT/b.log" 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/c" -- python3 scripts/conv_sweep.py "$LAYER" > "$OUT/c.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out = sys.argv[1]
for sub in "abc":
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            if "conv_igemm" not in k: continue
            print(k)
            for c, v in d.items():
                print("   %-28s %.5g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
tail -n 3 "$OUT"/a.log | cut -c1-300
