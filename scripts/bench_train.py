#!/usr/bin/env python3
"""Training-step benchmark (BASELINE configs[3]): ResNet-50, 640x480, 4 keypoints, nt = 11 frames
(1 labeled + 10 unlabeled), gm2=1 gm3=3, skeleton chain.  `bench_train.py [steps] [warm-up] [parity|f16]`: the parity tier (fp32-class
arithmetic) or the 16-bit tier of the training step (Trainer(tier="f16"), include/dgp_hip.h dgp_trainer_set_tier).  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgraphpose_amd.train import Trainer
from deepgraphpose_amd.loss import DGPHyper
from deepgraphpose_amd import dataset as D
from deepgraphpose_amd.arch import conv_macs_per_frame
from deepgraphpose_amd.synthetic import make_frames, make_weights

H, W, NJ, NT = 480, 640, 4, 11
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 2        # (the first pass has no predicted scales yet: its weight gradients run the fp32 tile)
tier = sys.argv[3] if len(sys.argv) > 3 else "parity"
repeats = int(sys.argv[4]) if len(sys.argv) > 4 else 1     # timed blocks of `steps`; ms_per_step is the FIRST block (the contract), the others are listed
rng = np.random.default_rng(0)
wts = make_weights(50, NJ, True, seed=0, head_std=0.05)
frames = torch.from_numpy(make_frames(NT, H, W, NJ, seed=0)).cuda()
jl = np.stack([rng.uniform(5, 55, (1, NJ)), rng.uniform(5, 75, (1, NJ))], -1)
vm, hm, vt = D.gen_idx_chunk(np.array([5]), np.setdiff1d(np.arange(NT), [5]), jl)
lt, lm = D.coord2map(jl, 60, 80, NJ, 17)
lmap, lmask = np.zeros((NT, 60, 80, 2 * NJ), np.float32), np.zeros((NT, 60, 80, 2 * NJ), np.float32)      # as the fit drivers build them
lmap[5], lmask[5] = lt[0], lm[0]
batch = dict(targets=jl, locref_map=lmap, locref_mask=lmask, visible_marker=vm, hidden_marker=hm, visible_marker_in_targets=vt)
S0 = np.zeros((3, NJ)); [S0.__setitem__((i, i), 1) or S0.__setitem__((i, i + 1), -1) for i in range(3)]
hy = DGPHyper(gm2=1, gm3=3)
tr = Trainer(50, NJ, H, W, max_frames=NT, tier=tier)
tr.load_weights(wts)
ws, ws_max = np.full(3, 10.0), np.full(3, 200.0)
for _ in range(warm):
    losses = tr.step(frames, batch, hy, S0, ws, ws_max, 2000.0, 50.0)
blocks = []
for _ in range(repeats):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = tr.step(frames, batch, hy, S0, ws, ws_max, 2000.0, 50.0)
    torch.cuda.synchronize()
    blocks.append((time.perf_counter() - t0) / steps)
dt = blocks[0]
fwd = 2.0 * conv_macs_per_frame(H, W, 50, NJ, True) * NT
# step-level roofline: forward + data-gradient + weight-gradient convolutions = 3 x the forward conv FLOPs (the stem has no data
# gradient: -1 %), every product as 3 fp16 MFMAs -> nominal ceiling 2500 / 3 TFLOP/s of algorithmic FLOPs.  Per-kernel times of the
# same command: profiles/r2_train_step_kernel_stats.txt (bash scripts/profile_train.sh).
ach = 3 * fwd / dt / 1e12
peak = 2500.0 if tier == "f16" else 833.3
print(json.dumps({"metric": "train_step", "ms_per_step": round(dt * 1e3, 2), "frames_per_s": round(NT / dt, 1),
                  "blocks_ms": [round(b * 1e3, 2) for b in blocks], "nt": NT, "dtype": "f16" if tier == "f16" else "f32", "tier": tier, "approx_tflops(3x fwd conv flops)": round(ach, 1),
                  "fast_passes": tr.fast_passes, "fast_redos": tr.fast_redos,
                  "roofline": {"bound": "mfma", "scope": "whole training step (forward + dgrad + wgrad convs, loss, clip, momentum)",
                               "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                               "peak_basis": ("dense f16 MFMA peak 2500 TFLOP/s (one MFMA per product in blocks 2-4; block1, root block and heads on the parity kernels)"
                                              if tier == "f16" else "dense f16 MFMA peak 2500 TFLOP/s / 3 partial products per fp32-class product"),
                               "algorithmic_gflop_per_step": round(3 * fwd / 1e9, 1)},
                  "loss": {k: round(v, 5) for k, v in losses.items()}}))
