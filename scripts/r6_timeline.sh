#!/bin/bash
# per-CU timeline of every conv launch of the bench workload (diagnostic build): scripts/r6_timeline.sh [parity|f16]
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/tl
for t in f16 parity; do
DGP_HIP_LIB=build_diag/libdgp_diag.so timeout 600 python scripts/diag_net.py $t 2>&1 | grep "diag split\|diag timeline" | tail -120 > gpurun_out/tl/tl_$t.txt
done
grep -B1 "timeline" gpurun_out/tl/tl_f16.txt | cut -c1-420 | tail -90
