#!/bin/bash
# the 256-row tile on the parity tier: layer tests, bit-identity of the network, per-launch A/B (profiles/r6_w64_h2_ab_layers.txt)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/w64h2
timeout 1500 python -m pytest tests/test_h2_gpu.py tests/test_h1_gpu.py -x -q -k "256_row" 2>&1 | tail -8 | tee gpurun_out/w64h2/tests.txt
timeout 1500 bash scripts/ab_envtier.sh DGP_W64 0 2 parity "block[234]" 2 2>&1 | tee gpurun_out/w64h2/ab.txt
