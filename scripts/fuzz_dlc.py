"""Random DLC (step-0) training-step configurations against the fp64 autograd oracle.  Usage: python scripts/fuzz_dlc.py [--sequence]
--sequence: ONE trainer per bodypart count taken through five frame sizes in a row, like fit_dlc's loader (pose_defaultdataset.py:131-196) -- the
trainer predicts this step's tensor scales from the previous step's ranges, so a size change must not leave stale state behind."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_train_gpu import _dlc_targets
from deepgraphpose_amd import synthetic
from deepgraphpose_amd.train import Trainer
from oracle import dgp_train_oracle as T
rng = np.random.default_rng(7)
bad = 0
SEQ = "--sequence" in sys.argv
tr = wts = None
for k in range(15 if SEQ else 10):
    if not SEQ or k % 5 == 0:
        nj = int(rng.choice([1, 2, 3, 5, 8]))
        wts = synthetic.make_weights(50, nj, True, seed=int(rng.integers(1 << 20)), head_std=0.05)
        tr = Trainer(50, nj, 64, 64, max_frames=1); tr.load_weights(wts)
    H, W = int(rng.integers(40, 140)), int(rng.integers(40, 140))
    P = T.make_params(wts, torch.float64)
    frames = synthetic.make_frames(1, H, W, nj, seed=k)
    if SEQ and k % 5 in (2, 4):
        frames = (frames // (6 if k % 5 == 2 else 1)).astype(np.uint8) if k % 5 == 2 else np.clip(frames.astype(np.int32) * 3, 0, 255).astype(np.uint8)
    sc, lmap, lmask = _dlc_targets(rng, H, W, nj)
    tr.set_input_size(H, W)
    losses = tr.forward_backward_dlc(torch.from_numpy(frames).cuda(), sc, lmap, lmask, locref_loss_weight=0.05)
    pred, loc = T.network(frames, P, 50, torch.float64)
    L = T.dlc_loss(pred, loc, torch.from_numpy(sc).double(), torch.from_numpy(lmap).double(), torch.from_numpy(lmask).double(), None, 0.05)
    L["total_loss"].backward()
    g = tr.get_grads()
    tot_ref = tot_err = 0.0; head = 0.0
    for name, t in P.items():
        if not t.requires_grad or t.grad is None: continue
        ref = t.grad.numpy(); d = g[name].reshape(ref.shape) - ref
        tot_ref += float((ref ** 2).sum()); tot_err += float((d ** 2).sum())
        if name.startswith("pose/"): head = max(head, np.linalg.norm(d.ravel()) / (np.linalg.norm(ref.ravel()) + 1e-30))
    e_loss = abs(losses["total_loss"] - float(L["total_loss"].detach())) / max(1.0, float(L["total_loss"].detach()))
    e_g = np.sqrt(tot_err / max(tot_ref, 1e-300))
    ok = e_loss < 2e-4 and e_g < 2e-2 and head < 1e-4
    bad += not ok
    print("%s nj %d %3d x %3d  loss %.2g grad %.2g heads %.2g" % ("ok " if ok else "BAD", nj, H, W, e_loss, e_g, head), flush=True)
print("failures:", bad)
