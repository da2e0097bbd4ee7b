"""Per-parameter comparison of one training step's gradients: the trainer against the oracle's fp64 and fp32 autograd (the case of tests/test_train_gpu.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np, torch
from test_train_gpu import _train_case, _oracle_grads
from deepgraphpose_amd.train import Trainer
from deepgraphpose_amd.loss import DGPHyper
batch, S0, wts, frames, ws, ws_max = _train_case(3)
hy = DGPHyper(gm2=1, gm3=3)
P64, L64 = _oracle_grads(wts, frames, batch, S0, ws, ws_max, hy, 300.0, 25.0, dtype=torch.float64)
P32, L32 = _oracle_grads(wts, frames, batch, S0, ws, ws_max, hy, 300.0, 25.0, dtype=torch.float32)
tr = Trainer(50, 3, 64, 96, max_frames=3); tr.load_weights(wts)
losses = tr.forward_backward(torch.from_numpy(frames).cuda(), batch, hy, S0, ws, ws_max, 300.0, 25.0)
g = tr.get_grads()
print(losses, float(L64["total_loss"]))
for k, t in P64.items():
    if not t.requires_grad: continue
    r64 = t.grad.numpy(); r32 = P32[k].grad.numpy().astype(np.float64); mine = g[k].reshape(r64.shape)
    d = lambda a, b: np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-30)
    e1, e2 = d(mine, r64), d(r32, r64)
    if e1 > 1e-4 or "block3/unit_6" in k or "conv1/weights" == k[-13:]:
        print("%-70s hip-vs-64 %.2e  torch32-vs-64 %.2e  |g| %.2e" % (k, e1, e2, np.linalg.norm(r64)))
