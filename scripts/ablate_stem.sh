#!/bin/bash
# Timing-only ablations of the root-block kernel's phases, no stamps (results of these builds are garbage): plain layer time of stem_pool_fused in
# builds with -DDGP_SX=<bits> (1 no input fetch, 2 no phase 1, 4 no MFMAs, 8 no phase-3 tile store, 16 no pooling reads, 32 no global stores).
# Usage: scripts/ablate_stem.sh <bits> ...
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/ablate_stem"; mkdir -p "$OUT" "$ROOT/build_diag"; cd "$ROOT"
for u in "$@"; do
  LIB=build_diag/libdgp_sx$u.so
  [ -f $LIB ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -DDGP_SX=$u -o $LIB deepgraphpose_amd/csrc/*.hip 2>/dev/null
  DGP_BENCH_ALLOW_OVERFLOW=1 DGP_HIP_LIB=$LIB timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --streams 1 --profile-steps 10 --no-strict-f32 \
      --layer-table "$OUT/lt_sx$u.tsv" > "$OUT/bench_sx$u.json" 2> "$OUT/bench_sx$u.err"
  echo "DGP_SX=$u  $(grep stem_pool_fused "$OUT/lt_sx$u.tsv" | awk -F'\t' '{print $4, "ms"}')  $(tail -1 "$OUT/bench_sx$u.err" | cut -c1-100)"
done
