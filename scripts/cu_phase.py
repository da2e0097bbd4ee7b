#!/usr/bin/env python3
"""Are the two workgroups a CU holds in phase?  From the diagnostic build's raw timeline dump (DGP_DIAG_DUMP=<file> with scripts/diag_net.py): for every launch,
per CU, the workgroups are assigned to two slots greedily; for every workgroup start, the phase of the OTHER slot's running workgroup (0 = it started at the same
time, 0.5 = it is half way through its life).  Prints the histogram of phases per launch shape.   python scripts/cu_phase.py dump.txt"""
import sys, collections
launches = []
for line in open(sys.argv[1]):
    if line.startswith("#"):
        launches.append((line[2:].strip(), []))
    else:
        a, b, k = line.split(); launches[-1][1].append((int(a), int(b), int(k)))
agg = collections.OrderedDict()
for name, wgs in launches:
    wgs = [w for w in wgs if w[1] > w[0]]
    if not wgs: continue
    span = max(w[1] for w in wgs) - min(w[0] for w in wgs)
    if span > 500000: continue                      # (supertile grids: padded blocks carry stale stamps)
    by = collections.defaultdict(list)
    for a, b, k in wgs: by[k].append((a, b))
    hist = [0] * 10
    for k, lst in by.items():
        lst.sort()
        slots = [[], []]
        for a, b in lst:
            s = 0 if (not slots[0] or slots[0][-1][1] <= a) else 1
            if s == 1 and slots[1] and slots[1][-1][1] > a: s = 0 if slots[0][-1][1] < slots[1][-1][1] else 1
            slots[s].append((a, b))
        for s in (0, 1):
            o = slots[1 - s]
            for a, b in slots[s]:
                for a2, b2 in o:
                    if a2 <= a < b2:
                        ph = (a - a2) / max(1, b2 - a2); hist[min(9, int(ph * 10))] += 1; break
    h = agg.setdefault(name, [0] * 10)
    for i in range(10): h[i] += hist[i]
print("phase of the co-resident workgroup when a workgroup starts (share of starts per decile of the other's life)")
for name, h in agg.items():
    n = max(1, sum(h))
    print("%-44s %s" % (name, " ".join("%4.2f" % (x / n) for x in h)))
