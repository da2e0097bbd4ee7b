"""Whole-net scoremap error vs the CPU oracle over a list of frame sizes (one frame each).  Usage: python scripts/big_frame_sizes.py H W [H W ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgraphpose_amd.engine import DGPNet
from deepgraphpose_amd.synthetic import make_frames, make_weights
from oracle import dgp_oracle as O
a = [int(v) for v in sys.argv[1:]]
nj = int(os.environ.get("NJ", 2))
wts = make_weights(50, nj, False, seed=9, head_std=0.05)
for H, W in zip(a[0::2], a[1::2]):
    frames = make_frames(1, H, W, nj, seed=10)
    ref = O.infer(frames, wts, 50, 8.0, 1.0, 1)
    net = DGPNet(50, nj, H, W, max_batch=1); net.load_weights(wts)
    sc = torch.empty((1, net.out_h, net.out_w, nj), device="cuda")
    feat = None
    mu, conf, idx = net.infer(torch.from_numpy(frames).cuda(), scmap_out=sc)
    sc = sc.cpu().numpy()
    bad = np.argwhere(np.abs(sc - ref["scmap"]) > 1e-2)
    print("%4d x %4d  feat %3d x %3d  scmap max abs err %.3g  bad cells %d %s" % (H, W, net.feat_h, net.feat_w, np.abs(sc - ref["scmap"]).max(), len(bad),
          (bad[:3].tolist(), bad[-3:].tolist()) if len(bad) else ""), flush=True)
    del net
