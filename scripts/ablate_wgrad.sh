#!/bin/bash
# timing-only ablations of wgrad_h3p (-DDGP_WX=n builds; results are garbage, only the times mean anything)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for x in 0 1 2 3 4 5; do
  lib=build_diag/libdgp_wx$x.so
  [ -f $lib ] || continue
  echo "== DGP_WX=$x"
  DGP_HIP_LIB=$lib python3 scripts/diag_wgrad.py 2>&1 | grep "ms per" | tr '\n' ' '; echo
done
