#!/bin/bash
# s_memtime stamps (-DDGP_DIAG build) of one 3x3 layer with and without the halo walk.  Usage: diag_halo.sh [N H W Cin Cout k stride rate]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p build_diag
[ -f build_diag/libdgp_diag.so ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -DDGP_DIAG -o build_diag/libdgp_diag.so deepgraphpose_amd/csrc/*.hip 2>/dev/null
SHAPE="${*:-32 30 40 512 512 3 1 2}"
for h in 0 1; do
  echo "== DGP_HALO=$h"
  DGP_HALO=$h DGP_HIP_LIB=build_diag/libdgp_diag.so python3 scripts/h2_conv_once.py $SHAPE 3 2>&1 | grep "^\[diag" | tail -3 | cut -c1-400
done
