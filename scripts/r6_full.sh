#!/bin/bash
# the round's full verification on one box: pytest -m gpu, __graft_entry__.smoke(), the default bench.py line
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6full
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r6full/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/r6full/smoke.txt
timeout 900 python bench.py 2> gpurun_out/r6full/bench.err | tee gpurun_out/r6full/bench.json | cut -c1-600
