"""Trained-like BN statistics (synthetic.make_stress_weights): soft-arg-max error of every conv mode AND of the CPU-fp32 oracle
against the fp64 anchor (oracle.dgp_oracle.infer(dtype=float64)).
Usage: python scripts/stress_modes.py [H W seed ...] -- runs itself once per mode in child processes (the modes are read once)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = [int(v) for v in sys.argv[1:]] or [480, 640, 3]
H, W, seeds = args[0], args[1], args[2:] or [3]
if os.environ.get("STRESS_CHILD"):
    import torch
    from deepgraphpose_amd.engine import DGPNet
    for seed in seeds:
        d = dict(np.load("/tmp/stress_case_%d.npz" % seed))
        frames = d.pop("frames"); r32 = d.pop("ref_mu"); ref_idx = d.pop("ref_idx"); s32 = d.pop("ref_sc"); r64 = d.pop("ref_mu64"); s64 = d.pop("ref_sc64")
        net = DGPNet(50, 4, H, W, max_batch=frames.shape[0]); net.load_weights(d)
        sc = torch.empty((frames.shape[0], net.out_h, net.out_w, 4), device="cuda")
        mu, conf, idx = net.infer(torch.from_numpy(frames).cuda(), scmap_out=sc)
        mu = mu.cpu().numpy().astype(np.float64); sc = sc.cpu().numpy().astype(np.float64)
        print("%-24s seed %d  px vs fp64 %.3g  vs oracle32 %.3g | oracle32 vs fp64 %.3g | idx equal %s | scmap rel vs fp64 %.3g (oracle32: %.3g)" % (
              os.environ["STRESS_CHILD"], seed, np.abs(mu - r64).max() * 8.0, np.abs(mu - r32).max() * 8.0, np.abs(r32 - r64).max() * 8.0,
              np.array_equal(idx.cpu().numpy(), ref_idx), np.abs(sc - s64).max() / np.abs(s64).max(), np.abs(s32 - s64).max() / np.abs(s64).max()), flush=True)
        del net
    sys.exit(0)
from deepgraphpose_amd.synthetic import make_frames, make_stress_weights
from oracle import dgp_oracle as O
for seed in seeds:
    kw = dict(a_lo=float(os.environ.get("A_LO", -6)), n_outliers=int(os.environ.get("N_OUT", 3)))
    wts = make_stress_weights(50, 4, False, seed=seed, **{k: v for k, v in kw.items() if k in make_stress_weights.__code__.co_varnames})
    frames = make_frames(2, H, W, 4, seed=seed + 1)
    s_ref, _ = O.pose_heads(O.resnet_features(frames, wts, 50), wts, False)
    wts["pose/part_pred/block4/weights"] = (wts["pose/part_pred/block4/weights"] * np.float32(float(os.environ.get("TARGET_STD", 3.0)) / s_ref.std())).astype(np.float32)
    ref = O.infer(frames, wts, 50, 8.0, 1.0, 1)
    r64 = O.infer(frames, wts, 50, 8.0, 1.0, 1, dtype=np.float64)
    np.savez("/tmp/stress_case_%d.npz" % seed, frames=frames, ref_mu=ref["mu"], ref_idx=ref["idx"], ref_sc=ref["scmap"], ref_mu64=r64["mu"], ref_sc64=r64["scmap"], **wts)
modes = (("H2 (default)", {}), ("DGP_H2=0", {"DGP_H2": "0"}), ("DGP_CONV_MODE=bf16x6", {"DGP_CONV_MODE": "bf16x6"}),
         ("DGP_CONV_MODE=f32", {"DGP_CONV_MODE": "f32"}), ("DGP_CHAIN=0", {"DGP_CHAIN": "0"}))
only = os.environ.get("STRESS_MODES")
for name, env in modes:
    if only and name not in only.split(","):
        continue
    subprocess.call([sys.executable, os.path.abspath(__file__), str(H), str(W)] + [str(s) for s in seeds], env=dict(os.environ, STRESS_CHILD=name, **env))
