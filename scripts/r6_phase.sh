#!/bin/bash
# raw per-workgroup timelines of every conv launch (DGP_DIAG_DUMP) -> scripts/cu_phase.py: are the two workgroups of a CU in phase?
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/tl; rm -f gpurun_out/tl/dump_*.txt
for t in parity f16; do
DGP_DIAG_DUMP=gpurun_out/tl/dump_$t.txt DGP_HIP_LIB=build_diag/libdgp_diag.so timeout 600 python scripts/diag_net.py $t > /dev/null 2>&1
python scripts/cu_phase.py gpurun_out/tl/dump_$t.txt > gpurun_out/tl/phase_$t.txt; cat gpurun_out/tl/phase_$t.txt
gzip -f gpurun_out/tl/dump_$t.txt
done
