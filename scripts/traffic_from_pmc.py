#!/usr/bin/env python3
"""HBM traffic of the conv launches from the rocprofv3 PMC passes of scripts/profile.sh.

usage: traffic_from_pmc.py <gpurun_out/prof_TAG> <profiles/traffic_TAG.json>
FETCH_SIZE / WRITE_SIZE are collected in SEPARATE passes (they do not fit one pass; MI355X_MICROARCH.md, rocprofv3 PMC slots) and are
reported in KiB; on gfx950 FETCH_SIZE tallies a wide coalesced read at half its bytes (128-byte requests counted as 64) -> x2.
Per kernel: launches per step and bytes per launch; `conv_hbm_bytes_per_step` sums every kernel that is part of the conv stack
(implicit-GEMM kernels, the fused root block, the grid-tail fixups, the head gather)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

prof, out = sys.argv[1], sys.argv[2]


def load(sub, counter):
    f = glob.glob(os.path.join(prof, sub, "**", "*counter_collection.csv"), recursive=True)
    acc, cnt = defaultdict(float), defaultdict(int)
    if not f:
        return acc, cnt
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("void ", "").replace("dgp::", "").split("(")[0].replace(" ", "")
        acc[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return acc, cnt


fetch, nf = load("pmc_fetch", "FETCH_SIZE")
write, nw = load("pmc_write", "WRITE_SIZE")
# each pass is normalised by ITS OWN number of forward passes: one h2_range_check launch per forward (the calibration call runs two:
# the layer-by-layer pass and the chained pass that follows it); engines without H2: one soft-argmax launch per step
steps_f = max(nf.get("h2_range_check_kernel", 0) or nf.get("soft_argmax_kernel", 0), 1)
steps_w = max(nw.get("h2_range_check_kernel", 0) or nw.get("soft_argmax_kernel", 0), 1)
steps = steps_f
CONV = ("conv_igemm", "chain_kernel", "unit_kernel", "stem_pool_fused", "tail_fixup", "head_gather", "maxpool3x3s2", "preprocess_u8", "reduce_slabs")
per_kernel, tot_fetch, tot_write = {}, 0.0, 0.0
for k in sorted(set(fetch) | set(write)):
    if not any(c in k for c in CONV):
        continue
    fb, wb = fetch.get(k, 0.0) * 1024.0 * 2.0, write.get(k, 0.0) * 1024.0
    fpl, wpl = fb / max(nf.get(k, 0), 1), wb / max(nw.get(k, 0), 1)
    per_kernel[k] = {"launches_per_step": nf.get(k, 0) / steps_f, "hbm_bytes_per_launch": fpl + wpl,
                     "fetch_bytes_per_launch_corrected": fpl, "write_bytes_per_launch": wpl}
    targs = k[k.index("<") + 1:k.rindex(">")].split(",") if "<" in k and k.startswith("conv_igemm_split_ls") else []
    if len(targs) >= 13 and targs[12] == "true":      # template H1: the 16-bit tier's instances of the cell kernels
        per_kernel[k]["h1"] = True
    tot_fetch += fb / steps_f
    tot_write += wb / steps_w
res = {"conv_hbm_bytes_per_step": tot_fetch + tot_write, "fetch_bytes_per_step_corrected": tot_fetch,
       "write_bytes_per_step": tot_write, "fetch_correction": 2.0, "steps_in_profile": steps, "per_kernel": per_kernel,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around bench.py, scripts/profile.sh"}
json.dump(res, open(out, "w"), indent=1)
print("conv stack: %.2f GB per step (fetch %.2f corrected over %d steps, write %.2f over %d steps)" % (
    res["conv_hbm_bytes_per_step"] / 1e9, tot_fetch / 1e9, steps_f, tot_write / 1e9, steps_w))
