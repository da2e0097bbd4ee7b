#!/bin/bash
# Timing-only ablations of the dominant kernel's operand feed (results are garbage): libraries built with -DDGP_FEEDX=1 (no A rows
# loaded after the prologue), 2 (no weight cells), 3 (neither) in scripts/micro/ next to the real one, in the 16-bit tier (a third of the
# MFMAs) and in the default tier, layer by layer.  Columns: block3/unit_3 conv1 conv2 conv3, block4/unit_2 conv1 conv2 conv3 (ms).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for mode in f16 f16x3; do
  for x in 0 1 2 3; do
    lib=deepgraphpose_amd/libdgp_hip.so
    [ $x != 0 ] && lib=scripts/micro/libdgp_feedx$x.so
    [ -f $lib ] || continue
    echo "mode $mode feedx $x: $(DGP_HIP_LIB=$ROOT/$lib DGP_CHAIN=0 DGP_CONV_MODE=$mode python3 scripts/ablate_feed.py 2>&1 | tail -1)"
  done
done
