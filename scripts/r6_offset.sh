#!/bin/bash
# does a phase offset between the two batches in flight change the throughput? (scripts/bench_tier.py --offset-ms; EXPERIMENTS.md R6 (8b))
cd ${GRAFT_REPO_ROOT:-.}
for T in parity f16; do for O in 0 1 2 3.3 5 0; do
echo "$T offset $O ms: $(timeout 300 python scripts/bench_tier.py $T --steps 200 --offset-ms $O 2>&1 | grep -E 'two streams' | sed 's/.*: //')"
done; done
