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
    return n[:70]


st = find("trace", "*kernel_stats.csv")
if st:
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    rows = list(csv.DictReader(open(st)))
    for r in rows[:12]:
        print("%-72s calls %6s  total %10.3f ms  avg %9.3f us  %6s%%" % (
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


# ---- traffic json for bench.py's roofline.traffic: HBM bytes of the conv launches of ONE step -------------
# FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch; gfx950 FETCH_SIZE counts 128-B requests as 64 B for
# wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM section).  Separate --pmc passes.
import json
steps = 7.0          # bench.py --steps 5 --warmup 2 in scripts/profile.sh
tot = {"fetch_kib": 0.0, "write_kib": 0.0}
per_kernel = defaultdict(lambda: {"fetch_kib": 0.0, "write_kib": 0.0, "launches": 0})
for sub, key, cname in (("pmc_fetch", "fetch_kib", "FETCH_SIZE"), ("pmc_write", "write_kib", "WRITE_SIZE")):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        if "conv_igemm" in r["Kernel_Name"] and r["Counter_Name"] == cname:
            tot[key] += float(r["Counter_Value"])
            k = r["Kernel_Name"].replace("void ", "").replace("dgp::", "").split("(")[0].replace(" ", "")
            per_kernel[k][key] += float(r["Counter_Value"])
            if key == "fetch_kib":
                per_kernel[k]["launches"] += 1
if tot["fetch_kib"] > 0:
    per_step = (2.0 * tot["fetch_kib"] + tot["write_kib"]) * 1024.0 / steps
    js = {"conv_hbm_bytes_per_step": per_step, "fetch_kib_raw_per_step": tot["fetch_kib"] / steps,
          "write_kib_per_step": tot["write_kib"] / steps, "fetch_correction": 2.0, "steps_in_profile": steps,
          "per_kernel": {k: {"launches_per_step": v["launches"] / steps,
                             "hbm_bytes_per_launch": (2.0 * v["fetch_kib"] + v["write_kib"]) * 1024.0 / max(v["launches"], 1)}
                         for k, v in per_kernel.items()},
          "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around bench.py"}
    json.dump(js, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print("\nconv HBM traffic per step: %.2f GB (fetch x2 corrected %.2f GB + write %.2f GB)" % (
        per_step / 1e9, 2 * tot["fetch_kib"] * 1024 / steps / 1e9, tot["write_kib"] * 1024 / steps / 1e9))
