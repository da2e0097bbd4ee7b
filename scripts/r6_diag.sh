#!/bin/bash
# diagnostic build's stamps of every conv launch of the 16-bit tier, 128-row tile (DGP_W64=0) and 256-row tile (DGP_W64=2)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/w64
for v in 0 2; do
DGP_W64=$v DGP_HIP_LIB=build_diag/libdgp_diag.so timeout 600 python scripts/diag_net.py f16 2>&1 | grep "diag" | tail -110 > gpurun_out/w64/diag_$v.txt
done
tail -n 100 gpurun_out/w64/diag_2.txt
