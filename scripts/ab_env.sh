#!/bin/bash
# A/B of one environment switch on the GPU box: bench.py on one stream per value, rows of the chain / unit kernels from the layer table.
# Usage: scripts/ab_env.sh <out_dir> <ENV_NAME> <value> [value ...]
OUT=$1; VAR=$2; shift 2
mkdir -p "$OUT"
for c in "$@"; do
  env $VAR=$c timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --streams 1 --profile-steps 10 --no-strict-f32 \
      --layer-table "$OUT/layers_${VAR}_$c.tsv" > "$OUT/bench_${VAR}_$c.log" 2>&1
  echo "== $VAR=$c: $(grep -h '^{' "$OUT/bench_${VAR}_$c.log" | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], "frames/s", d["ms_per_step"], "ms  frac", d["roofline"]["frac"])')"
  grep "|unit_c\||chain_c" "$OUT/layers_${VAR}_$c.tsv" | awk -F'\t' '{n=split($2,a,"|"); printf "   %-22s %8s ms %8s TF/s\n", a[n], $4, $5}'
done
