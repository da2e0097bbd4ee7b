#!/bin/bash
# A/B of the chain kernel's workgroup shapes on the GPU box: bench.py on one stream per DGP_CHAIN_CFG, chain rows of the layer table.
# Usage: scripts/ab_chain.sh <out_dir> [cfg ...]
OUT=${1:-gpurun_out/ab_chain}; shift
CFGS=${@:-0 1 2 3}
mkdir -p "$OUT"
for c in $CFGS; do
  DGP_CHAIN_CFG=$c timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --streams 1 --profile-steps 10 --no-strict-f32 \
      --layer-table "$OUT/layers_cfg$c.tsv" > "$OUT/bench_cfg$c.log" 2>&1
  echo "== cfg $c: $(grep -h '^{' "$OUT/bench_cfg$c.log" | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], "frames/s", d["ms_per_step"], "ms")')"
  grep chain_ "$OUT/layers_cfg$c.tsv" | awk -F'\t' '{n=split($2,a,"|"); printf "   %-22s %8s ms %8s TF/s\n", a[n], $4, $5}'
done
