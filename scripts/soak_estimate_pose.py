#!/usr/bin/env python3
"""Soak of the host pipeline of estimate_pose (round 6: ordered multi-thread staging, pinned ring and session kept between calls): many calls
of random length / batch size / tier on one snapshot, every one bit-identical to the first call of its (frames, batch size, tier); no thread
left behind, device memory flat.   python scripts/soak_estimate_pose.py [calls] [seed]"""
import os, sys, tempfile, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, yaml
from deepgraphpose_amd import weights_io
from deepgraphpose_amd.models import eval as E
from deepgraphpose_amd.synthetic import make_weights, make_frames

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
tmp = tempfile.mkdtemp()
proj = os.path.join(tmp, "proj"); train = os.path.join(proj, "dlc-models", "iteration-0", "DemoOct2-trainset95shuffle1", "train")
os.makedirs(train)
parts = ["a", "b", "c", "d"]
yaml.safe_dump(dict(Task="Demo", date="Oct2", iteration=0, TrainingFraction=[0.95], bodyparts=parts, skeleton=[], project_path=proj), open(os.path.join(proj, "config.yaml"), "w"))
yaml.safe_dump(dict(num_joints=4, all_joints_names=parts, net_type="resnet_50"), open(os.path.join(train, "pose_cfg.yaml"), "w"))
snap = weights_io.save_weights(os.path.join(train, "snapshot-step2-final--0"), make_weights(50, 4, False, seed=0, head_std=0.05))
cfgp = os.path.join(proj, "config.yaml")
base = make_frames(64, 128, 160, 4, seed=0)
seen, t0, frames_done = {}, time.perf_counter(), 0
m0 = None
for i in range(n_calls):
    T = int(rng.integers(1, 400)); bs = int(rng.choice([1, 3, 8, 16, 32])); tier = str(rng.choice(["parity", "f16"]))
    start = int(rng.integers(0, 64))
    fr = np.concatenate([base[start:], base[:start]] * (T // 64 + 1))[:T]
    out = E.estimate_pose(cfgp, snap, fr, os.path.join(tmp, "o%d" % i), save_pose=False, batch_size=bs, tier=tier)
    assert out["x"].shape == (T, 4) and np.isfinite(out["x"]).all() and np.isfinite(out["likelihoods"]).all()
    # frame t shows base[(start + t) % 64]: every frame of a video is calibrated on the video's FIRST batch, so compare whole calls only
    key = (T, bs, tier, start)
    if key in seen:
        assert all(np.array_equal(out[k], seen[key][k]) for k in ("x", "y", "likelihoods")), key
    seen[key] = out
    if i % 7 == 0:                                         # the same call again, at once: kept session, kept ring
        again = E.estimate_pose(cfgp, snap, fr, os.path.join(tmp, "o%d" % i), save_pose=False, batch_size=bs, tier=tier)
        assert all(np.array_equal(out[k], again[k]) for k in ("x", "y", "likelihoods")), ("repeat", key)
    frames_done += T
    torch.cuda.synchronize()
    if i == 5:
        m0 = torch.cuda.memory_allocated()
    if i % 50 == 49:
        print("call %d: device memory %.1f MB allocated, %.1f MB reserved" % (i + 1, torch.cuda.memory_allocated() / 1e6, torch.cuda.memory_reserved() / 1e6), flush=True)
alive = [t.name for t in threading.enumerate() if t is not threading.main_thread() and t.is_alive() and "stage" in (t.name or "")]
print("%d calls, %d frames in %.1f s; threads alive: %d; device memory since call 5: %+d bytes; reruns in the last call: %d"
      % (n_calls, frames_done, time.perf_counter() - t0, threading.active_count() - 1, torch.cuda.memory_allocated() - (m0 or 0), E.RUN_STATS["chunk_reruns"]))
assert threading.active_count() - 1 <= 2, [t.name for t in threading.enumerate()]
print("soak ok")
