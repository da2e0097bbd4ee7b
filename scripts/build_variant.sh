#!/bin/bash
# A/B helper: build_diag/libdgp_<name>.so = the in-tree objects with ONE source recompiled under extra flags.
# Usage: scripts/build_variant.sh <name> <source stem, e.g. dgp_kernels> <flags ...>      (run `python -m deepgraphpose_amd.build` first)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
N=$1; S=$2; shift 2
mkdir -p build_diag
B=deepgraphpose_amd/csrc/build
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC "$@" -c deepgraphpose_amd/csrc/$S.hip -o build_diag/${N}_$S.o
OBJS=""
for f in dgp_kernels dgp_ops dgp_chain dgp_loss dgp_net dgp_train; do
  if [ $f = $S ]; then OBJS="$OBJS build_diag/${N}_$S.o"; else OBJS="$OBJS $B/$f.hip.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o build_diag/libdgp_$N.so $OBJS
echo build_diag/libdgp_$N.so
