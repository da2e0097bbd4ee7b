#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/order
for t in f16 parity; do
  echo "== tier $t"
  timeout 900 bash scripts/ab_tier.sh deepgraphpose_amd/libdgp_hip.so build_diag/libdgp_$1.so $t "block[234]" 2 2>&1 | tail -42
done 2>&1 | tee gpurun_out/order/ab_$1.txt
