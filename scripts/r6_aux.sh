#!/bin/bash
# cache policy of the K loop's operand loads: base library vs -DDGP_A_AUX / -DDGP_B_AUX variants (results identical, timing differs); one stream + two streams, both tiers
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/aux; mkdir -p $OUT
for T in f16 parity; do
for L in base "$@" base; do
  LIB=deepgraphpose_amd/libdgp_hip.so; [ $L != base ] && LIB=build_diag/libdgp_$L.so
  DGP_HIP_LIB=$LIB timeout 300 python scripts/bench_tier.py $T --steps 60 --table $OUT/lt_${T}_$L.tsv > $OUT/bt_${T}_$L.txt 2>&1
  echo "$T $L: $(grep -E 'one stream|two streams' $OUT/bt_${T}_$L.txt | sed 's/.*: //' | tr '\n' '|')"
done
done
