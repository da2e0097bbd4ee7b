#!/bin/bash
# parity tier's root block: pool rows per tile (-DDGP_STEM_PH=5: conv1 tile aliases the planes, four barriers; 4 / 3: own buffers, two barriers), same box, alternating
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/stem
for r in 1 2 3; do for L in base ph5 ph3; do
  LIB=deepgraphpose_amd/libdgp_hip.so; [ $L != base ] && LIB=build_diag/libdgp_$L.so
  DGP_HIP_LIB=$LIB timeout 300 python scripts/bench_tier.py parity --steps 40 --table gpurun_out/stem/ph_$L.tsv > gpurun_out/stem/ph_$L.txt 2>&1
  echo "$L: stem $(grep stem_pool_fused gpurun_out/stem/ph_$L.tsv | cut -f3) us | $(grep -E 'one stream' gpurun_out/stem/ph_$L.txt | sed 's/.*: //')"
done; done
