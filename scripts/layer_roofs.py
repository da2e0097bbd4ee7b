#!/usr/bin/env python3
"""Per-layer roofs from a bench layer table (bench.py --layer-table): for every conv launch of a 32-frame step the achieved matrix-pipe
rate (algorithmic FLOPs / time) and the algorithmic byte rate (tensors in + out + residual + weights, once each / time), each as a
fraction of its peak.  Usage: layer_roofs.py <table.tsv> [elem_bytes=4] [mfma_peak_tflops=833.3]"""
import csv, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgraphpose_amd import arch
tab = sys.argv[1]; eb = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0; peak = float(sys.argv[3]) if len(sys.argv) > 3 else 833.3
B, H, W, NJ = 32, 480, 640, 4
rows = list(csv.DictReader(open(tab), delimiter="\t"))
for r in rows:
    if "us" not in r: r["us"] = float(r["avg_ms"]) * 1e3        # (bench.py --layer-table has avg_ms; the tier_f16 leg's table has us)
tot_t = sum(float(r["us"]) for r in rows)
print("%-52s %8s %8s %7s %9s %7s" % ("launch", "us", "TFLOP/s", "of pipe", "alg TB/s", "of HBM"))
for r in rows:
    name, kern = r["name"].split("|") if "|" in r["name"] else (r["name"], "")
    us = float(r["us"]); gf = float(r["gflop"])
    by = arch.launch_algorithmic_bytes(r["name"], H, W, 50, B, eb)
    tf = gf / (us * 1e-6) / 1e3
    tb = (by / (us * 1e-6) / 1e12) if by else float("nan")
    short = name.replace("conv:resnet_v1_50/", "").replace("bottleneck_v1/", "")
    print("%-52s %8.1f %8.1f %7.2f %9.2f %7.2f  %s" % (short[:52], us, tf, tf / peak, tb, tb / 8.0, kern))
print("sum %.1f us" % tot_t)
