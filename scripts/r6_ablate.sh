#!/bin/bash
# the 256-row tile's K loop piece by piece: timing-only -DDGP_X variants of the W64 loop (diagnostic build's stamps); profiles/r6_w64_stamps_ablation.txt
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/w64
for v in "$@"; do
echo "== $v"
DGP_W64=2 DGP_HIP_LIB=build_diag/libdgp_$v.so timeout 600 python scripts/diag_net.py f16 2>&1 | grep "diag w64" | tail -24 | sed 's/first barrier [0-9]* cyc, //' | sort | uniq | grep -E "tiles (600|2400) K-steps (32|24|72)"
done 2>&1 | tee gpurun_out/w64/ablate.txt
