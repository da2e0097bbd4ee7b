#!/bin/bash
# SQ counters of the unit / chain kernels (LDS conflicts, wait states) from a short bench run.  Usage: scripts/pmc_unit.sh <out_dir>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/${1:-gpurun_out/pmc_unit}
mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
BENCH="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --prewarm-seconds 0 --profile-steps 1 --streams 1 --no-strict-f32"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d "$OUT/a" -- $BENCH > "$OUT/a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d "$OUT/b" -- $BENCH > "$OUT/b.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, os, collections
for sub in ("a", "b"):
    f = glob.glob(os.path.join(sys.argv[1], sub, "**", "*counter_collection.csv"), recursive=True)
    if not f: print("no csv for", sub); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("void ", "").replace("dgp::", "")[:60]
        if not any(x in k for x in ("unit_kernel", "chain_kernel", "conv_igemm_split_ls<128, 128")): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(k)
        print("   " + "  ".join("%s %.4g" % (c, v / cnt[(k, c)]) for c, v in sorted(d.items())))
PY
