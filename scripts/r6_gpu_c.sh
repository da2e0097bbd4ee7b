#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6c
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "peaky" 2>&1 | grep -v "^$" | tail -8 | tee gpurun_out/r6c/peaky.txt
timeout 1500 python -m pytest tests/test_backward_layers_gpu.py -x -q -s -k "16_bit_tier" 2>&1 | grep -v "^$" | tail -8 | tee gpurun_out/r6c/train_fs.txt
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_h2_gpu.py -x -q -k "other_conv_modes or cell_kernels or halo" 2>&1 | tail -4 | tee gpurun_out/r6c/modes.txt
