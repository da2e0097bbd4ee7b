#!/bin/bash
# A/B helper: build_diag/libdgp_<name>.so = the library of git revision <rev> (sources checked out into a temporary worktree).
# Usage: scripts/build_rev.sh <rev> <name> [flags ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
REV=$1; N=$2; shift 2
WT=/tmp/dgp_wt_$N; rm -rf $WT; git worktree prune; git worktree add -f --detach $WT $REV > /dev/null 2>&1
mkdir -p build_diag $WT/o
for f in dgp_kernels dgp_ops dgp_chain dgp_loss dgp_net dgp_train; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC "$@" -c $WT/deepgraphpose_amd/csrc/$f.hip -o $WT/o/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o build_diag/libdgp_$N.so $WT/o/*.o
git worktree remove --force $WT
echo build_diag/libdgp_$N.so
