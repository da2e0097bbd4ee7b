#!/usr/bin/env python3
"""PCIe-inclusive throughput of estimate_pose (host frames -> pinned -> H2D -> dgp_infer) on a 640x480 frame stack."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, yaml
from deepgraphpose_amd import weights_io
from deepgraphpose_amd.models import eval as E
from deepgraphpose_amd.synthetic import make_weights, make_frames

T = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1024
TIER = sys.argv[sys.argv.index("--tier") + 1] if "--tier" in sys.argv else None      # None: DGP_EVAL_TIER or the parity tier
tmp = tempfile.mkdtemp()
proj = os.path.join(tmp, "proj"); train = os.path.join(proj, "dlc-models", "iteration-0", "DemoOct2-trainset95shuffle1", "train")
os.makedirs(train)
parts = ["a", "b", "c", "d"]
yaml.safe_dump(dict(Task="Demo", date="Oct2", iteration=0, TrainingFraction=[0.95], bodyparts=parts, skeleton=[], project_path=proj), open(os.path.join(proj, "config.yaml"), "w"))
yaml.safe_dump(dict(num_joints=4, all_joints_names=parts, net_type="resnet_50"), open(os.path.join(train, "pose_cfg.yaml"), "w"))
snap = weights_io.save_weights(os.path.join(train, "snapshot-step2-final--0"), make_weights(50, 4, False, seed=0, head_std=0.05))
base = make_frames(16, 480, 640, 4, seed=0)
frames = np.concatenate([base] * (T // 16))
E.estimate_pose(os.path.join(proj, "config.yaml"), snap, frames[:64], os.path.join(tmp, "warm"), save_pose=False, batch_size=32, tier=TIER)
import torch
torch.cuda.synchronize()
t0 = time.perf_counter()
out = E.estimate_pose(os.path.join(proj, "config.yaml"), snap, frames, os.path.join(tmp, "pred"), save_pose=False, batch_size=32, tier=TIER)
dt = time.perf_counter() - t0
if "--json" in sys.argv:
    import json
    setup = float(E.RUN_STATS.get("setup_s", 0.0))
    print(json.dumps({"frames_per_s": round(T / dt, 1), "frames": T, "seconds": round(dt, 3), "batch": 32, "tier": E.resolve_tier(TIER) or "parity",
                      "steady_frames_per_s": round(T / max(dt - setup, 1e-9), 1),      # without the call's fixed cost (snapshot -> engine), i.e. a long video's rate
                      "host_seconds": {k: round(float(v), 3) for k, v in E.RUN_STATS.items() if k.endswith("_s")},
                      "workload": "estimate_pose on a host array of %d 640x480x3 u8 frames (ResNet-50, 4 keypoints): engine set-up, decode thread, pinned "
                                  "staging (two threads), H2D on two copy streams, two engines, one D2H of the trajectory -- PCIe-inclusive, never `value`" % T}), flush=True)
print("estimate_pose on %d host frames (640x480x3 u8): %.1f frames/s incl. engine setup, pinned staging, H2D and the final D2H" % (T, T / dt))
