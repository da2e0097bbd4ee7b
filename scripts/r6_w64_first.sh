#!/bin/bash
# first GPU contact of the 256-row tile: layer tests under DGP_W64=2, bit-identity of the network, then the per-layer A/B
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/w64
timeout 1500 python -m pytest tests/test_h1_gpu.py -x -q -k "256_row or conv_on_h1" 2>&1 | tail -15 > gpurun_out/w64/tests.txt
cat gpurun_out/w64/tests.txt
timeout 1500 bash scripts/ab_envtier.sh DGP_W64 0 2 f16 "block[234]|part_pred" 2 2>&1 | tee gpurun_out/w64/ab.txt
