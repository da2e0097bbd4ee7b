#!/bin/bash
# s_memtime stamps of the unit kernels (-DDGP_DIAG build): one forward of the bench workload, the [diag unit ...] lines.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p build_diag
# DGP_UX (bit 0: no residual loads, bit 1: no X' stores) = timing-only ablations, results are garbage: libdgp_diag_ux<N>.so
LIB=build_diag/libdgp_diag${1:+_ux$1}.so
[ -f $LIB ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -DDGP_DIAG -DDGP_TUNING ${1:+-DDGP_UX=$1} -o $LIB deepgraphpose_amd/csrc/*.hip 2>/dev/null
DGP_HIP_LIB=$LIB python3 scripts/diag_net.py 2>&1 | grep "^\[diag unit\|^\[diag chain" | tail -8 | cut -c1-700
