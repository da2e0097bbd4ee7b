#!/usr/bin/env python3
"""Timeline of the training step from a rocprofv3 --kernel-trace (+ --memory-copy-trace) csv directory: per step the busy time (union of
kernel intervals), the gaps longer than a threshold with the kernels on either side, and the phase boundaries.
Usage: timeline_train.py <trace dir> [gap_us]"""
import csv, glob, os, sys
d = sys.argv[1]; thr = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(kt)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("dgp::", "").replace("(anonymous namespace)::", "")[:60]))
mc = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
for f in mc:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy:" + r.get("Direction", "")))
rows.sort()
# a step starts at preprocess_u8 of chain 1 (first of two launches per step)
starts = [i for i, r in enumerate(rows) if r[2].startswith("preprocess_u8")]
starts = starts[::2]
print("steps seen:", len(starts))
for si in range(len(starts) - 4, len(starts) - 1):
    a, b = starts[si], starts[si + 1]
    seg = rows[a:b]
    t0 = seg[0][0]; t1 = rows[b][0]
    busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
    gaps = []
    last_name = seg[0][2]
    for s, e, n in seg[1:] + [(t1, t1, "next step")]:
        if s > cur_e:
            busy += cur_e - cur_s
            if (s - cur_e) / 1e3 >= thr: gaps.append(((cur_e - t0) / 1e3, (s - cur_e) / 1e3, last_name, n))
            cur_s, cur_e = s, e; last_name = n
        else:
            if e > cur_e: cur_e = e; last_name = n
    print("step %d: %.3f ms wall, busy %.3f ms, idle %.3f ms, %d launches" % (si, (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(seg)))
    for at, g, p, n in gaps: print("    at %7.1f us  gap %6.1f us   after %-50s before %s" % (at, g, p, n))
    small = sum(1 for i in range(1, len(seg)) if 0 < seg[i][0] - max(x[1] for x in seg[max(0, i - 6):i]) < thr * 1e3)
