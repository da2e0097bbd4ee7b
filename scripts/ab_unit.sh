#!/bin/bash
# A/B of the unit kernel's shapes on the GPU box (needs a -DDGP_TUNING build).  Usage: scripts/ab_unit.sh <out_dir> [cfg ...]
OUT=${1:-gpurun_out/ab_unit}; shift
CFGS=${@:-0 1 2 3}
mkdir -p "$OUT"
n=0
for c in $CFGS; do
  n=$((n + 1))
  DGP_UNIT_CFG=$c timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --streams 1 --profile-steps 10 --no-strict-f32 \
      --layer-table "$OUT/layers_ucfg${c}_$n.tsv" > "$OUT/bench_ucfg${c}_$n.log" 2>&1
  echo "== unit cfg $c: $(grep -h '^{' "$OUT/bench_ucfg${c}_$n.log" | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], "frames/s", d["ms_per_step"], "ms")')"
  grep "unit_" "$OUT/layers_ucfg${c}_$n.tsv" | awk -F'\t' '{n=split($2,a,"|"); if (a[n] ~ /^unit_/) printf "   %-22s %8s ms %8s TF/s\n", a[n], $4, $5}'
  grep -i "traceback\|error\|mismatch" "$OUT/bench_ucfg${c}_$n.log" | head -3
done
