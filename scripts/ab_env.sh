#!/bin/bash
# A/B of one run-time switch on one box, alternating: one-stream bench per value, one row of the layer table.  Usage: scripts/ab_env.sh VAR v0 v1 <row-regex> [rounds]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/ab_env"; mkdir -p "$OUT"; cd "$ROOT"
VAR=$1; A=$2; B=$3; PAT=$4; R=${5:-3}
for r in $(seq 1 $R); do
  for v in $A $B; do
    env $VAR=$v timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --streams 1 --profile-steps 10 --no-strict-f32 \
        --layer-table "$OUT/lt_${v}_$r.tsv" > "$OUT/bench_${v}_$r.json" 2> "$OUT/bench_${v}_$r.err"
    python3 -c "
import json; d=json.load(open('$OUT/bench_${v}_$r.json')); print('$VAR=$v round $r:', d['value'], 'frames/s', d['ms_per_step'], 'ms   ', end='')"
    grep -E "$PAT" "$OUT/lt_${v}_$r.tsv" | awk -F'\t' '{n=split($2,a,"|"); printf "%s %s ms  ", a[n], $4}'; echo
  done
done
