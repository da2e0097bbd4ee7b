#!/bin/bash
# timing-only A/B: base library vs a -DDGP_X=<n> variant (results of the variant are garbage), 16-bit tier then parity tier, per-launch tables
cd ${GRAFT_REPO_ROOT:-.}
V=${1:-x10}; OUT=gpurun_out/ablate_b; mkdir -p $OUT
for T in f16 parity; do
for L in base $V; do
  LIB=deepgraphpose_amd/libdgp_hip.so; [ $L != base ] && LIB=build_diag/libdgp_$L.so
  DGP_HIP_LIB=$LIB timeout 300 python scripts/bench_tier.py $T --steps 60 --timing-only --table $OUT/lt_${T}_$L.tsv > $OUT/bt_${T}_$L.txt 2>&1
  echo "$T $L: $(grep -E 'one stream|two streams' $OUT/bt_${T}_$L.txt | sed 's/.*: //' | tr '\n' '|')"
done
python3 - $OUT $T $V <<'PY'
import sys, csv
out, T, V = sys.argv[1:4]
def tab(l): return [(x[0], float(x[2]), x[-1]) for x in list(csv.reader(open("%s/lt_%s_%s.tsv" % (out, T, l)), delimiter="\t"))[1:]]
a, b = tab("base"), tab(V)
sa = sb = 0
for (n, ta, k), (_, tb, _k) in zip(a, b):
    sa += ta; sb += tb
    print("%-60s %8.1f %8.1f %+6.1f %%  %s" % (n.replace("conv:resnet_v1_50/", "").replace("bottleneck_v1/", "")[-60:], ta, tb, (tb / ta - 1) * 100, k))
print("%-60s %8.1f %8.1f %+6.1f %%" % ("sum", sa, sb, (sb / sa - 1) * 100))
PY
done
