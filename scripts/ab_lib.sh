#!/bin/bash
# A/B of two builds of the library on one box: one-stream bench per library (alternating), per-layer times side by side.
# Usage: scripts/ab_lib.sh <libA.so> <libB.so> [row-regex] [rounds]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/ab_lib"; mkdir -p "$OUT"; cd "$ROOT"
A=$1; B=$2; PAT=${3:-.}; R=${4:-2}
for r in $(seq 1 $R); do
  for t in A B; do
    L=$A; [ $t = B ] && L=$B
    DGP_HIP_LIB=$L timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --streams 1 --profile-steps 10 --no-strict-f32 \
        --layer-table "$OUT/lt_${t}$r.tsv" > "$OUT/bench_${t}$r.json" 2> "$OUT/bench_${t}$r.err"
    python3 -c "
import json; d=json.load(open('$OUT/bench_${t}$r.json')); print('$t$r ($L):', d['value'], 'frames/s', d['ms_per_step'], 'ms  frac', d['roofline']['frac'])"
  done
done
python3 - "$OUT" "$PAT" $R <<'PY'
import sys, csv, re
out, pat, R = sys.argv[1], sys.argv[2], int(sys.argv[3])
def tab(t, r): return [(x[1], float(x[3])) for x in list(csv.reader(open("%s/lt_%s%d.tsv" % (out, t, r)), delimiter="\t"))[1:]]
a = [tab("A", r) for r in range(1, R + 1)]; b = [tab("B", r) for r in range(1, R + 1)]
sa = sb = 0.0
for i, (name, _) in enumerate(a[0]):
    ta = min(x[i][1] for x in a); tb = min(x[i][1] for x in b)
    if re.search(pat, name):
        sa += ta; sb += tb
        print("%-58s %8.4f %8.4f  %+5.1f %%" % (name.replace("conv:resnet_v1_50/", "").replace("bottleneck_v1/", "")[-58:], ta, tb, (tb / ta - 1) * 100))
print("%-58s %8.4f %8.4f  %+5.1f %%" % ("sum of the rows shown (min over rounds)", sa, sb, (sb / sa - 1) * 100))
PY
