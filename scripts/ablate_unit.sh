#!/bin/bash
# Timing-only ablations of the unit / chain kernels WITHOUT stamps (the results of these builds are garbage): plain layer times of builds with
# -DDGP_UX=<bits> (1 no residual loads, 2 no X' stores, 4 per-workgroup chunk rotation, 8 / 16 conv2 stage without LDS weight reads / MFMAs,
# 32 no halo DMA after the first tile).  Usage: scripts/ablate_unit.sh <bits> ...
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/ablate_unit"; mkdir -p "$OUT" "$ROOT/build_diag"; cd "$ROOT"
for u in "$@"; do
  LIB=build_diag/libdgp_ux$u.so
  [ -f $LIB ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -DDGP_TUNING -DDGP_UX=$u -o $LIB deepgraphpose_amd/csrc/*.hip 2>/dev/null
  DGP_BENCH_ALLOW_OVERFLOW=1 DGP_HIP_LIB=$LIB timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --streams 1 --profile-steps 10 --no-strict-f32 \
      --layer-table "$OUT/lt_ux$u.tsv" > "$OUT/bench_ux$u.json" 2> "$OUT/bench_ux$u.err"
  python3 - "$OUT/lt_ux$u.tsv" $u <<'PY'
import sys, csv
rows = list(csv.reader(open(sys.argv[1]), delimiter="\t"))[1:]
fused = [(r[1].split("|")[-1], float(r[3])) for r in rows if r[1].split("|")[-1].startswith(("unit_", "chain_"))]
print("DGP_UX=%-3s " % sys.argv[2] + "  ".join("%s %.4f" % (k.replace("chain_", "ch_").replace("unit_", "u_"), t) for k, t in fused))
PY
done
