#!/bin/bash
# Same-box A/B of one environment switch on the DEFAULT bench protocol (two streams), alternating values.  Usage: ab_bench.sh VAR v0 v1 [rounds]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$ROOT"
VAR=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq 1 $N); do
  for v in $A $B; do
    env $VAR=$v python bench.py --no-cpu-baseline --no-strict-f32 --sustain-seconds 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$VAR=$v', d['value'], 'frames/s', d['ms_per_step'], 'ms  frac', d['roofline']['frac'], ' sustained', d['sustained']['frames_per_s_per_gpu'], d['sustained']['sclk_mhz_under_load'])"
  done
done
