#!/bin/bash
# gpurun_out/refresh_<tag>/ -> profiles/ under the round's names (run in the build container after scripts/refresh_profiles.sh <tag> on the GPU box)
set -eu
TAG=${1:-r5}
ROOT=$(cd "$(dirname "$0")/.." && pwd); S=$ROOT/gpurun_out/refresh_$TAG; D=$ROOT/profiles
tail -1 "$S/bench_line.json" > "$D/${TAG}_bench_line.json"
cp "$S/kernel_stats.csv" "$D/${TAG}_kernel_stats.csv"
cp "$S/rocprofv3_summary.txt" "$D/${TAG}_rocprofv3_summary.txt"
cp "$S/traffic.json" "$D/traffic_${TAG}.json"
cp "$S/layer_table_hipevents.tsv" "$D/${TAG}_layer_table_hipevents.tsv"
cp "$S/layer_roofs.txt" "$D/${TAG}_layer_roofs.txt"
cp "$S/f16_kernel_stats.csv" "$D/${TAG}_f16_kernel_stats.csv"
cp "$S/f16_rocprofv3_summary.txt" "$D/${TAG}_f16_rocprofv3_summary.txt"
cp "$S/f16_traffic.json" "$D/traffic_${TAG}_f16.json"
cp "$S/train_step_kernel_stats.txt" "$D/${TAG}_train_step_kernel_stats.txt"
cp "$S/train_step_f16_kernel_stats.txt" "$D/${TAG}_train_step_f16_kernel_stats.txt"
cp "$S/train_step_f16_timeline.txt" "$D/${TAG}_train_step_f16_timeline.txt"
cp "$S/train_step_ab_tiers.txt" "$D/${TAG}_train_step_ab_tiers.txt"
cp "$S/traffic_per_layer_parity.txt" "$D/${TAG}_traffic_per_layer_parity.txt"
cp "$S/traffic_per_layer_f16.txt" "$D/${TAG}_traffic_per_layer_f16.txt"
cp "$S/f16_layer_table_hipevents.tsv" "$D/${TAG}_f16_layer_table_hipevents.tsv"
cp "$S/f16_layer_roofs.txt" "$D/${TAG}_f16_layer_roofs.txt"
python3 "$ROOT/scripts/readme_numbers.py"
ls -la "$D" | grep "${TAG}"
