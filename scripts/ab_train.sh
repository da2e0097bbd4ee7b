#!/bin/bash
# A/B of two builds of the library on one box, training step (scripts/bench_train.py), alternating.  Usage: scripts/ab_train.sh <libA.so> <libB.so> [tier] [rounds]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$ROOT"
A=$1; B=$2; T=${3:-f16}; R=${4:-2}
for r in $(seq 1 $R); do
  for L in $A $B; do
    echo -n "$L: "; DGP_HIP_LIB=$L timeout 300 python scripts/bench_train.py 50 8 $T 3 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['blocks_ms'], d['fast_redos'])"
  done
done
