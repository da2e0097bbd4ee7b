#!/usr/bin/env python3
"""Race check for the trainer's second stream: N forward/backward passes of the full-size step (640x480, 11 frames) from the same
weights must give the same gradients up to the order of the float atomics (~1e-6 relative); a stale read would show as a jump."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgraphpose_amd.train import Trainer
from deepgraphpose_amd.loss import DGPHyper
from deepgraphpose_amd import dataset as D
from deepgraphpose_amd.synthetic import make_frames, make_weights
H, W, NJ, NT = 480, 640, 4, 11
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(0)
wts = make_weights(50, NJ, True, seed=0, head_std=0.05)
frames = torch.from_numpy(make_frames(NT, H, W, NJ, seed=0)).cuda()
jl = np.stack([rng.uniform(5, 55, (1, NJ)), rng.uniform(5, 75, (1, NJ))], -1)
vm, hm, vt = D.gen_idx_chunk(np.array([5]), np.setdiff1d(np.arange(NT), [5]), jl)
lt, lm = D.coord2map(jl, 60, 80, NJ, 17)
lmap, lmask = np.zeros((NT, 60, 80, 2 * NJ)), np.zeros((NT, 60, 80, 2 * NJ))
lmap[5], lmask[5] = lt[0], lm[0]
batch = dict(targets=jl, locref_map=lmap, locref_mask=lmask, visible_marker=vm, hidden_marker=hm, visible_marker_in_targets=vt)
S0 = np.zeros((3, NJ)); [S0.__setitem__((i, i), 1) or S0.__setitem__((i, i + 1), -1) for i in range(3)]
hy = DGPHyper(gm2=1, gm3=3)
tr = Trainer(50, NJ, H, W, max_frames=NT)
tr.load_weights(wts)
ws, ws_max = np.full(3, 10.0), np.full(3, 200.0)
ref = None
worst = 0.0
for r in range(reps):
    tr.forward_backward(frames, batch, hy, S0, ws, ws_max, 2000.0, 50.0)
    g = tr.grads_tensor().clone()
    torch.cuda.synchronize()
    if ref is None:
        ref = g
        scale = float(ref.abs().max())
        continue
    d = float((g - ref).abs().max()) / scale
    worst = max(worst, d)
    if d > 1e-4:
        print("pass %d: max |g - g0| / max |g0| = %.3e  <-- suspicious" % (r, d))
print("reps %d  worst relative deviation from the first pass %.3e (atomics order alone: ~1e-6)" % (reps, worst))
sys.exit(1 if worst > 1e-4 else 0)
