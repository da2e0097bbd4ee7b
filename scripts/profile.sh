#!/bin/bash
# rocprofv3 passes for bench.py on the GPU box.  Usage: scripts/profile.sh <tag> [f16]
# 1) --kernel-trace --stats   2..4) separate --pmc passes (never combined with sys/hip traces).
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
BENCH="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --prewarm-seconds 0 --profile-steps 10 --streams 1 --no-strict-f32"
# second argument "f16": the 16-bit tier's child leg of bench.py (the same workload on H1 cells) instead of the parity tier's run
if [ "${2:-}" = "f16" ]; then BENCH="python3 bench.py --tier-f16-child --steps 10 --streams 1 --no-cpu-baseline"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_trace.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -- $BENCH > "$OUT/bench_pmc_mfma.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/bench_pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/bench_pmc_write.log" 2>&1
python3 scripts/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
python3 scripts/traffic_from_pmc.py "$OUT" "$OUT/traffic.json" >> "$OUT/summary.txt" 2>&1
cp "$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats.csv"
for f in "$OUT"/bench_*.log; do grep -h "^{" "$f" | cut -c1-400; done
find "$OUT" -name "*.csv" | head -20
du -sh "$OUT"
