#!/usr/bin/env python3
"""fast pass (H2 activations with predicted scales) against the plain pass: scoremaps, losses, gradients per tensor"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgraphpose_amd.train import Trainer
from deepgraphpose_amd.loss import DGPHyper
from deepgraphpose_amd import dataset as D
from deepgraphpose_amd.synthetic import make_frames, make_weights
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 640)
NJ, NT = 4, 11
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
out = []
for r in range(3):
    l = tr.forward_backward(frames, batch, hy, S0, ws, ws_max, 2000.0, 50.0)
    g = tr.get_grads()
    wsb, sc, lr = None, None, None
    out.append((l, g))
    print("pass", r, "fast redos", getattr(tr, "fast_redos", 0), {k: round(v, 6) for k, v in l.items()})
g0 = out[0][1]
for r in (1,):
    g = out[r][1]
    worst = []
    for k in g0:
        d = np.abs(g[k] - g0[k]).max() / (np.abs(g0[k]).max() + 1e-30)
        worst.append((d, k))
    worst.sort(reverse=True)
    print("pass", r, "worst tensors:")
    for d, k in worst[:12]:
        print("   %.3e  %s" % (d, k))
    bad = [k for d, k in worst if d > 1e-4]
    print("   tensors above 1e-4:", len(bad), "of", len(worst))
    order = list(g0.keys())
    print("   first bad in network order:", [k for k in order if k in bad][:6])
    gmax = max(float(np.abs(v).max()) for v in g0.values())
    print("   global max |g| %.3e; per tensor (network order): err relative to its own max | relative to the global max" % gmax)
    for k in order:
        e = float(np.abs(g[k] - g0[k]).max())
        print("   %-70s %.2e  %.2e  (max %.2e)" % (k[-70:], e / (float(np.abs(g0[k]).max()) + 1e-30), e / gmax, float(np.abs(g0[k]).max())))
