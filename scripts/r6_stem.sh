#!/bin/bash
# root-block work of round 6: the GPU suites that exercise it, then the per-launch time of both tiers
cd ${GRAFT_REPO_ROOT:-.}
timeout 2400 python -m pytest tests/test_h1_gpu.py tests/test_parity_gpu.py tests/test_boundary_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -3
for T in f16 parity; do
timeout 300 python scripts/bench_tier.py $T --steps 100 --table gpurun_out/stem_lt_$T.tsv 2>&1 | grep -E "stream|stem"
done
