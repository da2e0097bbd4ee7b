"""Trained-like BN statistics (synthetic.make_stress_weights): soft-arg-max error vs the CPU oracle under every conv mode.
Usage: python scripts/stress_modes.py [H W seed] -- runs itself once per mode in child processes (the modes are read once)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H, W, seed = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (480, 640, 3)))
if os.environ.get("STRESS_CHILD"):
    import torch
    from deepgraphpose_amd.engine import DGPNet
    d = dict(np.load("/tmp/stress_case.npz"))
    frames = d.pop("frames"); ref_mu = d.pop("ref_mu"); ref_idx = d.pop("ref_idx"); ref_sc = d.pop("ref_sc")
    net = DGPNet(50, 4, H, W, max_batch=frames.shape[0]); net.load_weights(d)
    sc = torch.empty((frames.shape[0], net.out_h, net.out_w, 4), device="cuda")
    mu, conf, idx = net.infer(torch.from_numpy(frames).cuda(), scmap_out=sc)
    print("%-28s px err %.3g  idx equal %s  scmap rel err %.3g" % (os.environ["STRESS_CHILD"], np.abs(mu.cpu().numpy() - ref_mu).max() * 8.0,
          np.array_equal(idx.cpu().numpy(), ref_idx), np.abs(sc.cpu().numpy() - ref_sc).max() / np.abs(ref_sc).max()), flush=True)
    sys.exit(0)
from deepgraphpose_amd.synthetic import make_frames, make_stress_weights
from oracle import dgp_oracle as O
kw = dict(a_lo=float(os.environ.get("A_LO", -6)), n_outliers=int(os.environ.get("N_OUT", 3)))
wts = make_stress_weights(50, 4, False, seed=seed, **{k: v for k, v in kw.items() if k in make_stress_weights.__code__.co_varnames})
frames = make_frames(2, H, W, 4, seed=seed + 1)
s_ref, _ = O.pose_heads(O.resnet_features(frames, wts, 50), wts, False)
wts["pose/part_pred/block4/weights"] = (wts["pose/part_pred/block4/weights"] * np.float32(float(os.environ.get("TARGET_STD", 3.0)) / s_ref.std())).astype(np.float32)
ref = O.infer(frames, wts, 50, 8.0, 1.0, 1)
np.savez("/tmp/stress_case.npz", frames=frames, ref_mu=ref["mu"], ref_idx=ref["idx"], ref_sc=ref["scmap"], **wts)
for name, env in (("H2 (default)", {}), ("DGP_H2=0", {"DGP_H2": "0"}), ("DGP_CONV_MODE=bf16x6", {"DGP_CONV_MODE": "bf16x6"}),
                  ("DGP_CONV_MODE=f32", {"DGP_CONV_MODE": "f32"}), ("DGP_CHAIN=0", {"DGP_CHAIN": "0"})):
    subprocess.call([sys.executable, os.path.abspath(__file__), str(H), str(W), str(seed)], env=dict(os.environ, STRESS_CHILD=name, **env))
