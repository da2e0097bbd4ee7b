#!/bin/bash
# A/B of one run-time switch on one box for one engine tier (scripts/bench_tier.py: one stream, two streams, per-launch table), alternating.
# Usage: scripts/ab_envtier.sh VAR v0 v1 [f16|parity] [row-regex] [rounds]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/ab_envtier"; mkdir -p "$OUT"; cd "$ROOT"
VAR=$1; A=$2; B=$3; T=${4:-f16}; PAT=${5:-.}; R=${6:-2}
for r in $(seq 1 $R); do
  for t in A B; do
    V=$A; [ $t = B ] && V=$B
    env $VAR=$V timeout 300 python scripts/bench_tier.py $T --steps 100 --table "$OUT/lt_${t}$r.tsv" > "$OUT/bt_${t}$r.txt" 2>&1
    echo "$t$r ($VAR=$V): $(grep -E 'one stream|two streams' "$OUT/bt_${t}$r.txt" | sed 's/.*: //' | tr '\n' '|')"
  done
done
python3 - "$OUT" "$PAT" $R <<'PY'
import sys, csv, re
out, pat, R = sys.argv[1], sys.argv[2], int(sys.argv[3])
def tab(t, r): return [(x[0], float(x[2]), x[-1]) for x in list(csv.reader(open("%s/lt_%s%d.tsv" % (out, t, r)), delimiter="\t"))[1:]]
a = [tab("A", r) for r in range(1, R + 1)]; b = [tab("B", r) for r in range(1, R + 1)]
sa = sb = 0.0
for i, (name, _, _k) in enumerate(a[0]):
    ta = min(x[i][1] for x in a); tb = min(x[i][1] for x in b)
    if re.search(pat, name):
        sa += ta; sb += tb
        print("%-62s %8.1f %8.1f  %+5.1f %%  %s" % (name.replace("conv:resnet_v1_50/", "").replace("bottleneck_v1/", "")[-62:], ta, tb, (tb / ta - 1) * 100, b[0][i][2]))
print("%-62s %8.1f %8.1f  %+5.1f %%" % ("sum of the rows shown (us, min over rounds)", sa, sb, (sb / sa - 1) * 100))
PY
