#!/bin/bash
# Everything profiles/ holds for a round, from ONE box.  Usage (on the GPU box): bash scripts/refresh_profiles.sh r5
# Outputs under gpurun_out/refresh_<tag>/ ; copy what is to be judged into profiles/ afterwards (scripts/collect_profiles.sh <tag>).
set -u
TAG=${1:-r5}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/refresh_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.log"
# parity tier: kernel trace + PMC passes of the bench command, per-launch table, every launch against both roofs
bash scripts/profile.sh "$TAG" > "$OUT/profile_sh.log" 2>&1
cp gpurun_out/prof_$TAG/kernel_stats.csv "$OUT/kernel_stats.csv"
cp gpurun_out/prof_$TAG/summary.txt "$OUT/rocprofv3_summary.txt"
cp gpurun_out/prof_$TAG/traffic.json "$OUT/traffic.json"
python3 bench.py --streams 1 --no-strict-f32 --no-cpu-baseline --layer-table "$OUT/layer_table_hipevents.tsv" > "$OUT/bench_one_stream.json" 2>/dev/null
python3 scripts/layer_roofs.py "$OUT/layer_table_hipevents.tsv" 4 833.3 > "$OUT/layer_roofs.txt" 2>&1
# 16-bit tier: the same passes on the tier_f16 leg
bash scripts/profile.sh "${TAG}_f16" f16 > "$OUT/profile_f16_sh.log" 2>&1
cp gpurun_out/prof_${TAG}_f16/kernel_stats.csv "$OUT/f16_kernel_stats.csv"
cp gpurun_out/prof_${TAG}_f16/summary.txt "$OUT/f16_rocprofv3_summary.txt"
cp gpurun_out/prof_${TAG}_f16/traffic.json "$OUT/f16_traffic.json"
# training step: per-kernel stats of both tiers (weight gradients on the chain's stream under the profiler), timeline of the 16-bit step
bash scripts/profile_train.sh > "$OUT/train_step_kernel_stats.txt" 2>&1
bash scripts/profile_train_f16.sh > "$OUT/train_step_f16_kernel_stats.txt" 2>&1
bash scripts/timeline_train.sh f16 15 > "$OUT/train_step_f16_timeline.txt" 2>&1
# memory-side traffic of EVERY launch of a step beside its algorithmic bytes (separate FETCH_SIZE / WRITE_SIZE passes), both tiers
bash scripts/traffic_per_layer.sh parity ${TAG}_parity > /dev/null 2>&1; cp gpurun_out/tpl_${TAG}_parity/per_layer.txt "$OUT/traffic_per_layer_parity.txt"
bash scripts/traffic_per_layer.sh f16 ${TAG}_f16 > /dev/null 2>&1; cp gpurun_out/tpl_${TAG}_f16/per_layer.txt "$OUT/traffic_per_layer_f16.txt"
python3 scripts/bench_tier.py f16 --steps 100 --table "$OUT/f16_layer_table_hipevents.tsv" > "$OUT/f16_bench_tier.txt" 2>&1
python3 scripts/layer_roofs.py "$OUT/f16_layer_table_hipevents.tsv" 2 2500 > "$OUT/f16_layer_roofs.txt" 2>&1
for i in 1 2; do python3 scripts/bench_train.py 50 8 f16 3 | cut -c1-140; python3 scripts/bench_train.py 50 8 parity 3 | cut -c1-140; done > "$OUT/train_step_ab_tiers.txt" 2>&1
ls -la "$OUT"
