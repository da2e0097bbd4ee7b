"""Random videos through estimate_pose (the A0 entry point: decode thread, pinned ring, chunked range checks, one D2H) against the oracle's
infer(): frame sizes 48-160 px, 1-45 frames, batch sizes 1-16, 1-6 bodyparts, ResNet-50 / 101, and a small DGP_EVAL_CHUNK_BYTES so that a
video spans several chunk rounds with ragged last batches.  Usage: python scripts/fuzz_estimate_pose.py [n] [seed]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, yaml
from deepgraphpose_amd import weights_io
from deepgraphpose_amd.models import eval as E
from deepgraphpose_amd.synthetic import make_weights, make_frames
from oracle import dgp_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = 0
for it in range(n):
    H, W = int(rng.integers(48, 161)), int(rng.integers(48, 161))
    T, bs, nj = int(rng.integers(1, 46)), int(rng.integers(1, 17)), int(rng.integers(1, 7))
    depth = 101 if rng.integers(0, 5) == 0 else 50
    chunk_batches = int(rng.integers(1, 5))
    tmp = tempfile.mkdtemp()
    proj = os.path.join(tmp, "proj"); train = os.path.join(proj, "dlc-models", "iteration-0", "DemoOct2-trainset95shuffle1", "train")
    os.makedirs(train)
    parts = ["p%d" % i for i in range(nj)]
    yaml.safe_dump(dict(Task="Demo", date="Oct2", iteration=0, TrainingFraction=[0.95], bodyparts=parts, skeleton=[], project_path=proj), open(os.path.join(proj, "config.yaml"), "w"))
    yaml.safe_dump(dict(num_joints=nj, all_joints_names=parts, net_type="resnet_%d" % depth), open(os.path.join(train, "pose_cfg.yaml"), "w"))
    wts = make_weights(depth, nj, False, seed=int(rng.integers(0, 1000)), head_std=0.05)
    snap = weights_io.save_weights(os.path.join(train, "snapshot-step2-final--0"), wts)
    frames = make_frames(T, H, W, nj, seed=int(rng.integers(0, 1000)))
    os.environ["DGP_EVAL_CHUNK_BATCHES"] = str(chunk_batches)
    out = E.estimate_pose(os.path.join(proj, "config.yaml"), snap, frames, os.path.join(tmp, "pred"), save_pose=False, batch_size=bs)
    ref = O.infer(frames, wts, depth, 8.0, 1.0, 1)
    dx = float(np.abs(out["x"] - ref["x"]).max()); dy = float(np.abs(out["y"] - ref["y"]).max()); dl = float(np.abs(out["likelihoods"] - ref["likelihoods"]).max())
    ok = out["x"].shape == (T, nj) and max(dx, dy) < 1e-3 and dl < 1e-4 and E.RUN_STATS["chunk_reruns"] == 0
    fails += 0 if ok else 1
    print("%s  R%d %3d x %3d  T %2d batch %2d nj %d chunk of %d batches (%d chunks)   |dx| %.1e |dy| %.1e px  lik %.1e"
          % ("ok  " if ok else "FAIL", depth, H, W, T, bs, nj, chunk_batches, E.RUN_STATS["chunks"], dx, dy, dl), flush=True)
print("failures: %d" % fails)
sys.exit(1 if fails else 0)
