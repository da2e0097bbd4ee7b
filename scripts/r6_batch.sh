#!/bin/bash
# does a smaller batch keep the layer-to-layer tensors in the memory-side cache?  per-launch tables at batch 32 / 16 / 8, time per 32 frames
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/batch; mkdir -p $OUT
for T in parity f16; do
for B in 32 16 8; do
  timeout 300 python scripts/bench_tier.py $T --batch $B --steps 60 --table $OUT/lt_${T}_$B.tsv > $OUT/bt_${T}_$B.txt 2>&1
  echo "$T batch $B: $(grep -E 'one stream|two streams' $OUT/bt_${T}_$B.txt | sed 's/.*: //' | tr '\n' '|')"
done
python3 - $OUT $T <<'PY'
import sys, csv
out, T = sys.argv[1:3]
def tab(b): return [(x[0], float(x[2])) for x in list(csv.reader(open("%s/lt_%s_%d.tsv" % (out, T, b)), delimiter="\t"))[1:]]
a, b, c = tab(32), tab(16), tab(8)
sa = sb = sc = 0
for (n, ta), (_, tb), (_, tc) in zip(a, b, c):
    sa += ta; sb += 2 * tb; sc += 4 * tc
    print("%-52s %8.1f %8.1f (%+5.1f%%) %8.1f (%+5.1f%%)" % (n.replace("conv:resnet_v1_50/", "").replace("bottleneck_v1/", "")[-52:], ta, 2 * tb, (2 * tb / ta - 1) * 100, 4 * tc, (4 * tc / ta - 1) * 100))
print("%-52s %8.1f %8.1f (%+5.1f%%) %8.1f (%+5.1f%%)" % ("sum (us per 32 frames)", sa, sb, (sb / sa - 1) * 100, sc, (sc / sa - 1) * 100))
PY
done
