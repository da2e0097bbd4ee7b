#!/bin/bash
# Memory-side traffic of EVERY launch of one forward step (rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE: separate passes, kernel-trace only)
# beside its algorithmic bytes.  Usage: scripts/traffic_per_layer.sh parity|f16 [out-tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${1:-parity}; TAG=${2:-$T}
OUT=$ROOT/gpurun_out/tpl_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/f" -- python3 scripts/run_forward.py $T 4 "$OUT/table.tsv" > "$OUT/f.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/w" -- python3 scripts/run_forward.py $T 4 > "$OUT/w.log" 2>&1
python3 scripts/traffic_per_layer.py "$OUT" $T | tee "$OUT/per_layer.txt"
