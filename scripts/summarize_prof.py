#!/usr/bin/env python3
"""Condense rocprofv3 csv outputs (kernel stats + PMC passes) into one text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    r = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return r[0] if r else None


def short(n):
    n = n.replace("dgp::", "")
    return n[:112]


st = find("trace", "*kernel_stats.csv")
if st:
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    rows = list(csv.DictReader(open(st)))
    for r in rows[:12]:
        print("%-114s calls %6s  total %10.3f ms  avg %9.3f us  %6s%%" % (
            short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
            r["Percentage"]))

for sub in ("pmc_mfma", "pmc_fetch", "pmc_write"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    print("\n== %s (per-dispatch average) ==" % sub)
    for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:8]:
        print(k)
        for c, v in d.items():
            print("    %-28s %.6g  (over %d dispatches)" % (c, v / cnt[(k, c)], cnt[(k, c)]))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
            n = cnt[(k, "SQ_VALU_MFMA_BUSY_CYCLES")]
            # MFMA_BUSY counts cycles summed over SIMDs (1024 on the chip); GUI_ACTIVE sums 8 XCDs
            util = (d["SQ_VALU_MFMA_BUSY_CYCLES"] / n) / ((d["GRBM_GUI_ACTIVE"] / n) / 8.0 * 1024.0)
            print("    MFMA utilisation ~ %.3f (MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs))" % util)


# (HBM traffic per step and per kernel: scripts/traffic_from_pmc.py, which normalises every pass by its own number of forward passes)
