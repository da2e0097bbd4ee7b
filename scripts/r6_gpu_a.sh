#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6a
timeout 1200 python -m pytest tests/test_train_gpu.py -x -q -k "stale_head_panels or tier_f16" 2>&1 | tail -15 | tee gpurun_out/r6a/train.txt
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "atrous" 2>&1 | tail -8 | tee gpurun_out/r6a/atrous.txt
