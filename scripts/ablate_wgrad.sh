#!/bin/bash
# timing-only ablations of wgrad_h3p (-DDGP_WX=n builds; results are garbage, only the times mean anything):
#   1 no global loads in the loop, 2 no MFMAs, 3 no transposed reads, 4 no split + LDS stores, 5 no mid-step barrier,
#   6 plain stores instead of the float atomics of the epilogue (all wgrad kernels)
# and the -DDGP_DIAG build for scripts/diag_wgrad.py (s_memtime stamps in wgrad_h3; run it with DGP_WGRAD_PIPE=0).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
mkdir -p build_diag
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for x in 1 2 3 4 5 6; do
  [ -f build_diag/libdgp_wx$x.so ] || $HIPCC -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -DDGP_WX=$x -o build_diag/libdgp_wx$x.so deepgraphpose_amd/csrc/*.hip 2>/dev/null
done
[ -f build_diag/libdgp_diag.so ] || $HIPCC -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -DDGP_DIAG -o build_diag/libdgp_diag.so deepgraphpose_amd/csrc/*.hip 2>/dev/null
echo "== default build"
python3 scripts/diag_wgrad.py 2>&1 | grep "ms per" | tr '\n' ' '; echo
for x in 1 2 3 4 5 6; do
  echo "== DGP_WX=$x"
  DGP_HIP_LIB=build_diag/libdgp_wx$x.so python3 scripts/diag_wgrad.py 2>&1 | grep "ms per" | tr '\n' ' '; echo
done
echo "== stamps (wgrad_h3, DGP_WGRAD_PIPE=0)"
DGP_WGRAD_PIPE=0 DGP_HIP_LIB=build_diag/libdgp_diag.so python3 scripts/diag_wgrad.py 2>&1 | grep "^\[diag" | awk '!seen[$4" "$6" "$8]++' | cut -c1-330
echo "== training step: atomics vs plain stores"
python3 scripts/bench_train.py 6 | cut -c1-80
DGP_HIP_LIB=build_diag/libdgp_wx6.so python3 scripts/bench_train.py 6 | cut -c1-80
