#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 1200 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "network_small or root_block or infer_640 or demo_frame" 2>&1 | tail -3
for T in f16 parity; do
timeout 300 python scripts/bench_tier.py $T --steps 100 --table gpurun_out/stem_lt_$T.tsv 2>&1 | grep -E "stream|stem"
done
