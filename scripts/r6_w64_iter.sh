#!/bin/bash
# one iteration on the 256-row tile: layer tests under DGP_W64=2 + bit-identity, stamps of the diagnostic build, per-layer A/B
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/w64
timeout 1500 python -m pytest tests/test_h1_gpu.py -x -q -k "256_row" 2>&1 | tail -5 > gpurun_out/w64/tests.txt
cat gpurun_out/w64/tests.txt
DGP_W64=2 DGP_HIP_LIB=build_diag/libdgp_diag.so timeout 600 python scripts/diag_net.py f16 2>&1 | grep "diag w64" | tail -24 | sort | uniq -c | sort -rn | head -12 > gpurun_out/w64/diag_2.txt
cat gpurun_out/w64/diag_2.txt
timeout 1500 bash scripts/ab_envtier.sh DGP_W64 0 2 f16 "block[34]" ${1:-2} 2>&1 | tee gpurun_out/w64/ab.txt
