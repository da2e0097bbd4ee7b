#!/bin/bash
# A/B of the halo walk (DGP_HALO=0/1) on one box: one-stream bench per value, the 3x3 rows of the layer table side by side.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/${1:-r4}"
mkdir -p "$OUT"; cd "$ROOT"
for h in 0 1; do
  DGP_HALO=$h timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --streams 1 --profile-steps 10 --no-strict-f32 \
      --layer-table "$OUT/lt_halo$h.tsv" > "$OUT/bench_halo$h.json" 2> "$OUT/bench_halo$h.err"
  python3 -c "
import json; d=json.load(open('$OUT/bench_halo$h.json')); print('DGP_HALO=$h', d['value'], 'frames/s', d['ms_per_step'], 'ms  frac', d['roofline']['frac'], 'dominant ms', d['roofline']['kernel_ms_per_step'])"
done
paste "$OUT/lt_halo0.tsv" "$OUT/lt_halo1.tsv" | awk -F'\t' '$2 ~ /conv2\|splith3_128x128/ {n=split($2,a,"/"); printf "%-8s %-8s %8s %8s  %+5.1f %%\n", a[2], a[3], $4, $9, ($9/$4-1)*100}'
