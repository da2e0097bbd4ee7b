#!/usr/bin/env python3
"""Timeline summary of one training step from a rocprofv3 --kernel-trace csv: per stream busy time, union, overlap, idle gaps."""
import csv, glob, sys, os
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are delimited by momentum_kernel (one per step); take the span between the last two
idx = [i for i, r in enumerate(rows) if "momentum_kernel" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"]); t1 = int(step[-1]["End_Timestamp"])
print("step span %.3f ms, %d launches" % ((t1 - t0) / 1e6, len(step)))
by = {}
for r in step:
    by.setdefault(r["Stream_Id"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for sid, v in by.items():
    busy = sum(e - s for s, e, _ in v)
    print("stream %s: %d launches, busy %.3f ms, first at %.3f ms, last end %.3f ms" % (sid, len(v), busy / 1e6, (v[0][0] - t0) / 1e6, (max(e for _, e, _ in v) - t0) / 1e6))
ev = sorted([(s, 1) for r in step for s in [int(r["Start_Timestamp"])]] + [(e, -1) for r in step for e in [int(r["End_Timestamp"])]])
cur, last, t_by = 0, t0, {0: 0, 1: 0, 2: 0}
for t, d in ev:
    t_by[min(cur, 2)] = t_by.get(min(cur, 2), 0) + (t - last)
    cur += d; last = t
print("time with 0 / 1 / >=2 kernels in flight: %.3f / %.3f / %.3f ms" % (t_by[0] / 1e6, t_by[1] / 1e6, t_by[2] / 1e6))
# where the second stream ends relative to the chain
names = lambda v: [n.split("(")[0][-24:] for _, _, n in v]
for sid, v in by.items():
    print("stream %s last kernels: %s" % (sid, ", ".join(names(v)[-3:])))

# idle gaps (no kernel in flight) longer than 30 us
ivs = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in step)
cur_end = ivs[0][1]
for s_, e_, n_ in ivs[1:]:
    if s_ - cur_end > 30000:
        print("idle %.3f ms before %-40s at %.3f ms" % ((s_ - cur_end) / 1e6, n_.replace("(anonymous namespace)::", "").split("(")[0][-40:], (s_ - t0) / 1e6))
    cur_end = max(cur_end, e_)
