#!/bin/bash
# per-layer times of bench.py under experimental builds (libdgp_hip_x<N>.so, -DDGP_X=N: wrong numerics, timing only)
cd ${GRAFT_REPO_ROOT:-.}
for x in base "$@"; do
  if [ $x = base ]; then unset DGP_HIP_LIB; else export DGP_HIP_LIB=deepgraphpose_amd/libdgp_hip_$x.so; fi
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --layer-table gpurun_out/lt_$x.tsv | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$x', d['value'], d['ms_per_step'])"
done
python3 - "$@" <<'PY'
import sys
xs = ["base"] + sys.argv[1:]
t = {}
for x in xs:
    for l in open("gpurun_out/lt_%s.tsv" % x).read().splitlines()[1:]:
        f = l.split("\t"); t.setdefault(f[0], {})[x] = (f[1], float(f[3]))
print("%-4s %-40s" % ("#", "layer") + "".join("%9s" % x for x in xs))
for k in sorted(t, key=int):
    name = t[k][xs[0]][0].replace("conv:resnet_v1_50/", "").replace("bottleneck_v1/", "")[:40]
    print("%-4s %-40s" % (k, name) + "".join("%9.4f" % t[k][x][1] for x in xs))
PY
