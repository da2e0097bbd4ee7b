#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6b
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_boundary_gpu.py -x -q -k "estimate_pose or pipeline or host or shard or fullsize or evaluate" 2>&1 | tail -12 | tee gpurun_out/r6b/tests.txt
for t in parity f16; do
timeout 300 python scripts/bench_pipeline.py 4096 --json --tier $t 2>&1 | grep "^{" | tee gpurun_out/r6b/pipe_$t.json
done
