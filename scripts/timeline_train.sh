#!/bin/bash
# rocprofv3 kernel + memory-copy trace of the training step in its production configuration (weight gradients on the second stream), then the gap analysis
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; T=${1:-f16}
OUT=$ROOT/gpurun_out/tl_$T; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace" -- python3 scripts/bench_train.py 10 4 $T > "$OUT/bench.log" 2>&1
grep "^{" "$OUT/bench.log" | cut -c1-120
python3 scripts/timeline_train.py "$OUT/trace" ${2:-15}
