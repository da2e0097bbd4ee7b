#!/bin/bash
# A/B of the number of loader waves in the unit / chain kernels (needs a -DDGP_TUNING build): one-stream bench per (DGP_UNIT_CFG, DGP_CHAIN_CFG) pair,
# fused rows of the layer table.  Usage: scripts/ab_loaders.sh "u,c" ...
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/ab_loaders"; mkdir -p "$OUT"; cd "$ROOT"
n=0
for pair in "$@"; do
  n=$((n + 1)); u=${pair%,*}; c=${pair#*,}
  DGP_UNIT_CFG=$u DGP_CHAIN_CFG=$c timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --streams 1 --profile-steps 10 --no-strict-f32 \
      --layer-table "$OUT/lt_${n}.tsv" > "$OUT/bench_${n}.json" 2> "$OUT/bench_${n}.err"
  python3 -c "
import json; d=json.load(open('$OUT/bench_${n}.json')); print('unit cfg $u chain cfg $c:', d['value'], 'frames/s', d['ms_per_step'], 'ms  max px err', d.get('accuracy', {}).get('px_max'))"
  awk -F'\t' '$2 ~ /unit_|chain_/ {n=split($2,a,"|"); printf "   %-22s %8s ms\n", a[n], $4}' "$OUT/lt_${n}.tsv" | paste - - - - 
done
