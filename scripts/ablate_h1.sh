#!/bin/bash
# Timing-only ablations of the 16-bit tier's conv kernel (results are garbage): per-launch ms of block3/unit_3 and block4/unit_2
# (conv1 conv2 conv3 each) under libraries built with -DDGP_FEEDX=1 (no A rows after the prologue), 2 (no weight cells), 3 (neither),
# -DDGP_X=6 (no B fragment reads from LDS in the 16x16x32 loop), -DDGP_X=7 (no MFMAs there).  Usage: scripts/ablate_h1.sh <dir with the .so files>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$ROOT"; D=${1:-ab}
for mode in f16 f16x3; do
  echo "mode $mode base: $(DGP_CHAIN=0 DGP_CONV_MODE=$mode python3 scripts/ablate_feed.py 2>&1 | tail -1)"
  for x in feedx1 feedx2 feedx3 x6 x7; do
    [ -f $D/$x.so ] || continue
    echo "mode $mode $x: $(DGP_HIP_LIB=$ROOT/$D/$x.so DGP_CHAIN=0 DGP_CONV_MODE=$mode python3 scripts/ablate_feed.py 2>&1 | tail -1)"
  done
done
