#!/bin/bash
# the trainer's small grids on 128 x 64 tiles: DGP_SMALL_GRID=<threshold in 128x128 tiles> on a -DDGP_TUNING build of dgp_kernels.hip
cd ${GRAFT_REPO_ROOT:-.}
for r in 1 2; do
for v in 0 "$@"; do
for T in f16 parity; do
echo "SMALL_GRID=$v $T: $(DGP_SMALL_GRID=$v DGP_HIP_LIB=build_diag/libdgp_tune.so timeout 300 python scripts/bench_train.py 50 8 $T 2>/dev/null | cut -c1-70)"
done; done; done
