#!/bin/bash
# Two ranks of a data-parallel training step on ONE GPU (gloo control plane), rank 0 under rocprofv3 (kernel + memory-copy trace): does the
# gradient exchange start while the backward pass is still running?   scripts/timeline_dp.sh [f16|parity]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; T=${1:-f16}
OUT=$ROOT/gpurun_out/tl_dp_$T; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
export WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29671 DGP_DIST_BACKEND=gloo LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
RANK=1 python3 scripts/dp_step_worker.py 6 $T > "$OUT/rank1.log" 2>&1 &
P1=$!
export RANK=0
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace" -- python3 scripts/dp_step_worker.py 6 $T > "$OUT/rank0.log" 2>&1
wait $P1
grep "^rank" "$OUT/rank0.log" "$OUT/rank1.log" | cut -c1-260
python3 scripts/timeline_dp.py "$OUT/trace"
