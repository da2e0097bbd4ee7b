#!/usr/bin/env python3
"""Per-launch times of a few layers of the layer-by-layer H2 engine under whatever library DGP_HIP_LIB names (timing-only ablation builds:
results are garbage, nothing is checked).  Prints 'name ms' rows for block3/unit_3 and block4/unit_2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepgraphpose_amd import engine
from deepgraphpose_amd.synthetic import make_frames, make_weights
B = 32
net = engine.DGPNet(50, 4, 480, 640, max_batch=B)
net.load_weights(make_weights(50, 4, False, seed=0, head_std=0.05))
fr = torch.from_numpy(make_frames(8, 480, 640, 4, seed=100)).cuda().repeat(4, 1, 1, 1).contiguous()
out = torch.zeros((B, 4, 5), device="cuda")
for _ in range(4):
    net.infer_packed(fr, out, 1.0, 1)
torch.cuda.synchronize()
net.profile_begin(6)
for _ in range(6):
    net.infer_packed(fr, out, 1.0, 1)
torch.cuda.synchronize()
n, rows = net.profile_end()
sel = [r for r in rows if "block3/unit_3" in r[0] or "block4/unit_2" in r[0]]
print(" ".join("%.4f" % r[2] for r in sel), "| total %.3f" % sum(r[2] for r in rows))
