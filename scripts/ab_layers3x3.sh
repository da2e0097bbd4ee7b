#!/bin/bash
# A/B of one switch on one box: one-stream bench per value, rows of the layer table matching a pattern side by side.
# Usage: ab_layers3x3.sh VAR v0 v1 [row-regex]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/r4"; VAR=$1; A=$2; B=$3; PAT=${4:-conv2\|splith3_128x128}
mkdir -p "$OUT"; cd "$ROOT"
tag() { echo "$1" | tr -c 'A-Za-z0-9_.\n' '_'; }
TA=$(tag "$A"); TB=$(tag "$B")
for h in $A $B; do
  T=$(tag "$h")
  env $VAR=$h timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --streams 1 --profile-steps 10 --no-strict-f32 \
      --layer-table "$OUT/lt_${VAR}_$T.tsv" > "$OUT/bench_${VAR}_$T.json" 2> "$OUT/bench_${VAR}_$T.err"
  python3 -c "
import json; d=json.load(open('$OUT/bench_${VAR}_$T.json')); print('$VAR=$h', d['value'], 'frames/s', d['ms_per_step'], 'ms  frac', d['roofline']['frac'], 'dominant ms', d['roofline']['kernel_ms_per_step'])"
done
paste "$OUT/lt_${VAR}_$TA.tsv" "$OUT/lt_${VAR}_$TB.tsv" | awk -F'\t' -v pat="$PAT" '$2 ~ pat {n=split($2,a,"/"); printf "%-8s %-8s %-28s %8s %8s  %+5.1f %%\n", a[2], a[3], substr(a[n],1,28), $4, $9, ($9/$4-1)*100}'
