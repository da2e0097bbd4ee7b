#!/bin/bash
# phases of the 16-bit tier's root block (two workgroups per CU): timing-only -DDGP_SX=<bits> builds of dgp_ops.hip (scripts/build_variant.sh sx<bits> dgp_ops -DDGP_SX=<bits>)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/stem
for u in base "$@" base; do
  LIB=deepgraphpose_amd/libdgp_hip.so; [ $u != base ] && LIB=build_diag/libdgp_sx$u.so
  DGP_HIP_LIB=$LIB timeout 300 python scripts/bench_tier.py f16 --steps 20 --timing-only --table gpurun_out/stem/lt_$u.tsv > gpurun_out/stem/bt_$u.txt 2>&1
  echo "DGP_SX=$u  $(grep stem_pool_fused gpurun_out/stem/lt_$u.tsv | cut -f3) us"
done
