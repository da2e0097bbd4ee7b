#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6d
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q -k "data_parallel or two_ranks or tier_f16 or step or fit" 2>&1 | tail -12 | tee gpurun_out/r6d/dp.txt
