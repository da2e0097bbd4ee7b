#!/usr/bin/env python3
"""Why a conv launch of the 16-bit tier sits where it does: the diagnostic build's in-kernel stamps (scripts/diag_net.py f16 on a -DDGP_DIAG
library) turned into one row per DISTINCT launch shape of the bench workload (ResNet-50, 640x480, batch 32) -- grid against the 512
resident slots (two 128 x 128 workgroups per CU), cycles of a tile by phase (compute wave 0), and what bounds the launch.
    python scripts/launch_causes.py gpurun_out/w64/diag_0.txt [layer_table.tsv] > profiles/r6_f16_launch_causes.txt"""
import re
import sys
from collections import OrderedDict

NAMES = {  # (tiles, K-steps) of the bench workload's launches -> layer(s)
    (1200, 18): "block2 conv2 (3x3, halo walk) x3", (300, 18): "block2/unit_4 conv2 (3x3 stride 2, per-tap)",
    (4800, 6): "block2/unit_1 conv3+shortcut (K 128+256)", (1200, 8): "block2/unit_2 conv1 (K 512)",
    (600, 36): "block3 conv2 (3x3, halo walk) x6", (2400, 12): "block3/unit_1 conv3+shortcut (K 256+512)", (600, 16): "block3 conv1 (K 1024) x5",
    (2400, 4): "block3 conv3 (K 256) x5", (1200, 16): "block4/unit_1 conv1 (K 1024)", (1200, 72): "block4 conv2 (3x3 rate 2, halo walk) x3",
    (4800, 24): "block4/unit_1 conv3+shortcut (K 512+1024)", (4864, 24): "block4/unit_1 conv3+shortcut (K 512+1024, supertile grid)", (1200, 32): "block4 conv1 (K 2048) x2", (4800, 8): "block4 conv3 (K 512) x2",
    (4800, 1): "block1 conv1 (K 64, 128x64 tile)", (1200, 9): "block1/unit_3 conv2 (3x3 stride 2, 128x64 tile)", (300, 32): "heads: pointwise GEMM (K 2048, 128x64 tile, fp32 out)",
}
pat = re.compile(r"\[diag split (\d+)x(\d+) .*?\] tiles (\d+) K-steps (\d+) \| compute wave 0: first barrier (\d+) cyc, epilogue (\d+) \| per K-step: "
                 r"ldsread\+mfma (\d+) barrier-wait (\d+) \|\| loader wave: .*?load-issue (\d+) barrier-wait (\d+)")
rows = OrderedDict()
for line in open(sys.argv[1]):
    m = pat.search(line)
    if not m:
        continue
    bm, bn, T, K, pro, epi, mf, bw, li, lbw = map(int, m.groups())
    if mf == 0:
        continue                                  # (32 x 64 wave tiles: the stamps of the loop are not taken)
    rows.setdefault((T, K, bm, bn), []).append((pro, epi, mf, bw, li, lbw))
print("# one row per distinct launch shape of the 16-bit tier's conv stack (bench workload; diagnostic build, compute wave 0 / a loader wave, cycles).")
print("# slots = 512 (two 128 x 128 workgroups per CU).  ideal = K-steps x 1024 cycles (32 MFMAs per wave and step, two waves per SIMD).")
print("%-52s %6s %6s %5s | %7s %8s %8s %7s | %5s %5s | %s" % ("launch", "tiles", "rounds", "K", "prolog", "K loop", "of it wait", "epilog", "MFMA", "tail", "what bounds it"))
for (T, K, bm, bn), v in rows.items():
    n = len(v)
    pro, epi, mf, bw, li, lbw = (sum(x[i] for x in v) / n for i in range(6))
    loop = K * (mf + bw)
    tile = pro + loop + epi
    ideal = K * 1024.0
    rounds = T / 512.0
    full, frac = int(rounds), rounds - int(rounds)
    # two co-resident workgroups overlap one's prologue / epilogue with the other's K loop: the pipe's share of a tile pair
    mfma_share = min(1.0, 2 * ideal / tile) if tile > 0 else 0.0
    tail_eff = rounds / (full + (1 if frac > 0 else 0))
    if K <= 8:
        why = "epilogue: %d cycles of residual + store per tile vs %d of K loop -> HBM-bound output stream" % (epi, loop)
    elif bw > 0.4 * mf:
        why = "loaders: a wave waits %d cycles per step for operands (loader issue %d)%s" % (bw, li, "; grid tail %.0f %%" % (100 * tail_eff) if tail_eff < 0.8 else "")
    elif tail_eff < 0.8:
        why = "grid tail: %d tiles on 512 slots = %.2f rounds (%.0f %% of the last round idle)" % (T, rounds, 100 * (1 - frac) if frac else 0)
    else:
        why = "matrix pipe + prologue / epilogue exposure (%.0f %% of a tile outside the K loop)" % (100 * (pro + epi) / tile)
    print("%-52s %6d %6.2f %5d | %7.0f %8.0f %8.0f %7.0f | %5.2f %5.2f | %s" % (NAMES.get((T, K), "%dx%d tile" % (bm, bn)), T, rounds, K, pro, loop, K * bw, epi,
                                                                               mfma_share, tail_eff, why))
