#!/bin/bash
# A/B of the epilogue's non-temporal policy (DGP_EPI_NT: 2 default, 3 loads only, 4 stores only, 0 off) on one box, one stream.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r4
for v in 2 3 4 0 2 3; do
  DGP_EPI_NT=$v python bench.py --steps 12 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --streams 1 --profile-steps 10 --no-strict-f32 --layer-table gpurun_out/r4/lt_nt$v.tsv 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('DGP_EPI_NT=$v', d['value'], d['ms_per_step'])"
  awk -F'\t' '$2 ~ /block3\/unit_[3]\/.*(conv3|conv1)|block4\/unit_[23]\/.*(conv3|conv1)/ {n=split($2,a,"/"); printf "   %s %s %s\n", a[2], a[3], $4}' gpurun_out/r4/lt_nt$v.tsv | tr '\n' ' '; echo
done
