#!/usr/bin/env python3
"""G of an inner unit after DGP_BWD_STOP units, plain pass vs fast pass (DGP_BWD_DUMP writes it)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["DGP_BWD_DUMP"] = "/tmp/gdump.bin"
from deepgraphpose_amd.train import Trainer
from deepgraphpose_amd.loss import DGPHyper
from deepgraphpose_amd import dataset as D
from deepgraphpose_amd.synthetic import make_frames, make_weights
H, W, NJ, NT = 480, 640, 4, 11
oh, ow = H // 8, W // 8
rng = np.random.default_rng(0)
wts = make_weights(50, NJ, True, seed=0, head_std=0.05)
frames = torch.from_numpy(make_frames(NT, H, W, NJ, seed=0)).cuda()
jl = np.stack([rng.uniform(5, oh - 5, (1, NJ)), rng.uniform(5, ow - 5, (1, NJ))], -1)
vm, hm, vt = D.gen_idx_chunk(np.array([5]), np.setdiff1d(np.arange(NT), [5]), jl)
lt, lm = D.coord2map(jl, oh, ow, NJ, 17)
lmap, lmask = np.zeros((NT, oh, ow, 2 * NJ)), np.zeros((NT, oh, ow, 2 * NJ))
lmap[5], lmask[5] = lt[0], lm[0]
batch = dict(targets=jl, locref_map=lmap, locref_mask=lmask, visible_marker=vm, hidden_marker=hm, visible_marker_in_targets=vt)
S0 = np.zeros((3, NJ)); [S0.__setitem__((i, i), 1) or S0.__setitem__((i, i + 1), -1) for i in range(3)]
hy = DGPHyper(gm2=1, gm3=3)
tr = Trainer(50, NJ, H, W, max_frames=NT)
tr.load_weights(wts)
ws, ws_max = np.full(3, 10.0), np.full(3, 200.0)
dumps = []
for r in range(2):
    tr.forward_backward(frames, batch, hy, S0, ws, ws_max, 2000.0, 50.0)
    torch.cuda.synchronize()
    dumps.append(np.fromfile("/tmp/gdump.bin", dtype=np.float32).copy())
a, b = dumps
print("stop", os.environ.get("DGP_BWD_STOP"), "n", a.size, "max|a|", np.abs(a).max(), "max|a-b|/max|a|", np.abs(a - b).max() / np.abs(a).max(),
      "frac differing > 1e-4 max:", float((np.abs(a - b) > 1e-4 * np.abs(a).max()).mean()), "nonzero frac a/b", float((a != 0).mean()), float((b != 0).mean()))

if os.environ.get("DGP_BWD_STOP") == "1":
    A = a.reshape(NT, 30, 40, -1); B = b.reshape(NT, 30, 40, -1)
    bad = np.abs(A - B) > 1e-4 * np.abs(a).max()
    print("bad by channel mod 8:", [round(float(bad[..., k::8].mean()), 4) for k in range(8)])
    print("bad by frame:", [round(float(bad[n].mean()), 4) for n in range(NT)])
    print("bad by row (frame 0):", [round(float(bad[0, r].mean()), 3) for r in range(30)])
    print("bad by col (frame 0):", [round(float(bad[0, :, c].mean()), 3) for c in range(40)])
    pm = bad.reshape(-1, bad.shape[-1]).mean(1)
    print("bad per pixel, first 64 px:", [round(float(v), 2) for v in pm[:64]])
    cm = bad.reshape(-1, bad.shape[-1]).mean(0)
    print("bad per channel, first 64:", [round(float(v), 2) for v in cm[:64]])
