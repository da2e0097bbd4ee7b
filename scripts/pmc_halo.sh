#!/bin/bash
# SQ counters + kernel time of one 3x3 layer with and without the halo walk.  Usage: pmc_halo.sh [N H W Cin Cout k stride rate]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/pmc_halo"
rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
SHAPE="${*:-32 30 40 512 512 3 1 2}"
for h in 0 1; do
  export DGP_HALO=$h
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/h$h/t" -- python3 scripts/h2_conv_once.py $SHAPE 8 > "$OUT/h$h.t.log" 2>&1
  i=0
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/h$h/p$i" -- python3 scripts/h2_conv_once.py $SHAPE 3 > "$OUT/h$h.p$i.log" 2>&1
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out = sys.argv[1]
res = {}
for h in "01":
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, "h" + h, "p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_igemm_split_ls" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for f in glob.glob(os.path.join(out, "h" + h, "t", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_igemm_split_ls" in r["Name"]:
                acc["avg_ns"] = [float(r["AverageNs"]), 1]
    res[h] = {k: v[0] / max(v[1], 1) for k, v in acc.items()}
print("%-28s %14s %14s %8s" % ("counter (per dispatch)", "HALO=0", "HALO=1", "ratio"))
for k in sorted(set(res["0"]) | set(res["1"])):
    a, b = res["0"].get(k, float("nan")), res["1"].get(k, float("nan"))
    print("%-28s %14.6g %14.6g %8.3f" % (k, a, b, b / a if a else float("nan")))
PY
