#!/bin/bash
# Everything profiles/ holds for a round, from ONE box.  Usage (on the GPU box): bash scripts/refresh_profiles.sh r3
# Outputs under gpurun_out/refresh_<tag>/ ; copy what is to be judged into profiles/ afterwards.
set -u
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/refresh_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.log"
bash scripts/profile.sh "$TAG" > "$OUT/profile_sh.log" 2>&1
cp gpurun_out/prof_$TAG/kernel_stats.csv "$OUT/kernel_stats.csv"
cp gpurun_out/prof_$TAG/summary.txt "$OUT/rocprofv3_summary.txt"
cp gpurun_out/prof_$TAG/traffic.json "$OUT/traffic.json"
python3 bench.py --streams 1 --no-strict-f32 --no-cpu-baseline --layer-table "$OUT/layer_table_hipevents.tsv" > "$OUT/bench_one_stream.json" 2>/dev/null
DGP_CHAIN=0 DGP_EPI_NT=0 python3 bench.py --streams 1 --no-strict-f32 --no-cpu-baseline --layer-table "$OUT/layer_table_hipevents_nochain.tsv" > "$OUT/bench_one_stream_nochain.json" 2>/dev/null
bash scripts/profile_train.sh > "$OUT/train_step_kernel_stats.txt" 2>&1
python3 scripts/group_conv_trace.py gpurun_out/prof_train 14 > "$OUT/train_step_conv_by_grid.txt" 2>&1
for i in 1 2 3; do python3 scripts/bench_train.py 100 | cut -c1-70; DGP_WGRAD_DMA=0 python3 scripts/bench_train.py 100 | cut -c1-70; done > "$OUT/train_step_ab_dma.txt" 2>&1
python3 scripts/bench_r101.py > "$OUT/r101_line.json" 2>/dev/null
ls -la "$OUT"
