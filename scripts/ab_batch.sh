#!/bin/bash
# Per-layer time PER FRAME at different batch sizes, one stream (does the front of the network gain from tensors that fit the Infinity Cache?).
# Usage: scripts/ab_batch.sh [batch ...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/ab_batch"; mkdir -p "$OUT"; cd "$ROOT"
BS=${@:-8 16 32}
for b in $BS; do
  timeout 300 python bench.py --batch $b --steps 12 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --streams 1 --profile-steps 10 --no-strict-f32 \
      --layer-table "$OUT/lt_b$b.tsv" > "$OUT/bench_b$b.json" 2> "$OUT/bench_b$b.err"
  python3 -c "
import json; d=json.load(open('$OUT/bench_b$b.json')); print('batch $b:', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')"
done
python3 - "$OUT" $BS <<'PY'
import sys, csv
out, bs = sys.argv[1], [int(b) for b in sys.argv[2:]]
tabs = {}
for b in bs:
    rows = list(csv.reader(open("%s/lt_b%d.tsv" % (out, b)), delimiter="\t"))[1:]
    tabs[b] = [(r[1], float(r[3])) for r in rows]
names = [n for n, _ in tabs[bs[-1]]]
print("%-64s" % "layer (us per frame)" + "".join("  b=%-6d" % b for b in bs))
groups = {}
for i, n in enumerate(names):
    line = "%-64s" % n.replace("conv:resnet_v1_50/", "").replace("bottleneck_v1/", "")[:64]
    for b in bs:
        t = dict(tabs[b]).get(n)
        line += "  %8.2f" % (t * 1e3 / b) if t is not None else "       n/a"
        g = n.split("/")[1] if "/" in n else n
        if t is not None: groups.setdefault(g, {}).setdefault(b, 0.0); groups[g][b] += t * 1e3 / b
    print(line)
print()
for g, d in groups.items():
    print("%-64s" % ("sum " + g) + "".join("  %8.2f" % d.get(b, float("nan")) for b in bs))
PY
