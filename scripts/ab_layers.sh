#!/bin/bash
# per-layer A/B of env switches on one box.  Usage: ab_layers.sh "VAR=val" ["VAR2=val" ...]   (first column: defaults)
cd ${GRAFT_REPO_ROOT:-.}
names="base"
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --layer-table gpurun_out/lt_base.tsv | tail -1 | cut -c1-60
i=0
for kv in "$@"; do
  i=$((i+1)); names="$names v$i"
  env $kv python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --layer-table gpurun_out/lt_v$i.tsv | tail -1 | cut -c1-60
done
python3 - $names <<'PY'
import sys
xs = sys.argv[1:]
t = {}
for x in xs:
    for l in open("gpurun_out/lt_%s.tsv" % x).read().splitlines()[1:]:
        f = l.split("\t"); t.setdefault(f[0], {})[x] = (f[1], float(f[3]))
print("%-4s %-40s" % ("#", "layer") + "".join("%9s" % x for x in xs))
tot = {x: 0.0 for x in xs}
for k in sorted(t, key=int):
    name = t[k][xs[0]][0].replace("conv:resnet_v1_50/", "").replace("bottleneck_v1/", "")[:40]
    print("%-4s %-40s" % (k, name) + "".join("%9.4f" % t[k][x][1] for x in xs))
    for x in xs: tot[x] += t[k][x][1]
print("%-45s" % "total" + "".join("%9.4f" % tot[x] for x in xs))
PY
