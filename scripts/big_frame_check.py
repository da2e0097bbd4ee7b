"""One frame at a size whose scoremap exceeds the LDS map of the soft-argmax, under the engine's A/B switches, against the CPU oracle.
Usage: python scripts/big_frame_check.py [H W]  (runs itself once per switch set in child processes)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H, W = (int(v) for v in (sys.argv[1:3] if len(sys.argv) >= 3 else (1616, 1600)))
if os.environ.get("BIG_CHILD"):
    import torch
    from deepgraphpose_amd.engine import DGPNet
    d = dict(np.load("/tmp/big_case.npz"))
    frames = d.pop("frames"); ref_mu = d.pop("ref_mu"); ref_sc = d.pop("ref_sc")
    net = DGPNet(50, 2, H, W, max_batch=1); net.load_weights(d)
    sc = torch.empty((1, net.out_h, net.out_w, 2), device="cuda")
    mu, conf, idx = net.infer(torch.from_numpy(frames).cuda(), scmap_out=sc)
    sc = sc.cpu().numpy()
    print("%-40s px err %.3g  scmap max abs err %.3g (max |ref| %.3g) finite %s  range_status %s" % (
        os.environ["BIG_CHILD"], np.abs(mu.cpu().numpy() - ref_mu).max() * 8.0, np.abs(sc - ref_sc).max(), np.abs(ref_sc).max(),
        np.isfinite(sc).all(), net.range_status()), flush=True)
    sys.exit(0)
from deepgraphpose_amd.synthetic import make_frames, make_weights
from oracle import dgp_oracle as O
wts = make_weights(50, 2, False, seed=9, head_std=0.05)
frames = make_frames(1, H, W, 2, seed=10)
ref = O.infer(frames, wts, 50, 8.0, 1.0, 1)
np.savez("/tmp/big_case.npz", frames=frames, ref_mu=ref["mu"], ref_sc=ref["scmap"], **wts)
for name, env in (("default", {}), ("DGP_HALO=0", {"DGP_HALO": "0"}), ("DGP_CHAIN=0", {"DGP_CHAIN": "0"}), ("DGP_CHAIN=0 DGP_HALO=0", {"DGP_CHAIN": "0", "DGP_HALO": "0"}),
                  ("DGP_H2=0", {"DGP_H2": "0"}), ("DGP_CONV_MODE=f32", {"DGP_CONV_MODE": "f32"}), ("DGP_STEM_FUSED=0", {"DGP_STEM_FUSED": "0"}),
                  ("DGP_HEAD_PW=0", {"DGP_HEAD_PW": "0"})):
    subprocess.call([sys.executable, os.path.abspath(__file__), str(H), str(W)], env=dict(os.environ, BIG_CHILD=name, **env))
