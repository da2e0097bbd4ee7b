"""Diagnose the nj=20 / dense-skeleton loss gradient at tiny map sizes (8 x 12): per-term comparison vs the fp64 autograd oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_train_gpu import _make_loss_case
from test_backward_layers_gpu import _dense_skeleton, _loss_cfg
from deepgraphpose_amd.loss import dgp_loss_fwd_bwd, DGPHyper
from oracle import dgp_train_oracle as T

nj, nt, H, W = 20, 3, 8, 12
rng = np.random.default_rng(101)
batch, _ = _make_loss_case(rng, nt, H, W, nj, 1, 0.1, 2)
S0 = _dense_skeleton(nj)
ws, ws_max = rng.uniform(5, 20, 190), rng.uniform(10, 40, 190)
pred = (rng.standard_normal((nt, H, W, nj)) * 0.3).astype(np.float32)
loc = (rng.standard_normal((nt, H, W, 2 * nj)) * 0.3).astype(np.float32)
for name, hy, S, w_, wm_ in (("all", DGPHyper(gm2=1, gm3=3), S0, ws, ws_max), ("no clique", DGPHyper(gm2=1, gm3=3), np.zeros((0, nj)), np.zeros(0), np.zeros(0)),
                             ("gm 0 0 + clique", DGPHyper(gm2=0, gm3=0), S0, ws, ws_max), ("gm 0 0 no clique", DGPHyper(gm2=0, gm3=0), np.zeros((0, nj)), np.zeros(0), np.zeros(0)),
                             ("clique 2 limbs", DGPHyper(gm2=0, gm3=0), S0[:2], ws[:2], ws_max[:2]), ("clique big ws_max", DGPHyper(gm2=0, gm3=0), S0, ws, ws_max * 100)):
    pt = torch.tensor(pred, dtype=torch.float64, requires_grad=True)
    lt = torch.tensor(loc, dtype=torch.float64, requires_grad=True)
    L = T.dgp_loss(pt, lt, batch, _loss_cfg(hy, nj, S, w_, wm_, 300.0, 25.0))
    L["total_loss"].backward()
    losses, dpred, dloc, mu = dgp_loss_fwd_bwd(torch.from_numpy(pred).cuda(), torch.from_numpy(loc).cuda(), batch, hy, S, w_, wm_, 300.0, 25.0)
    gp = pt.grad.numpy(); d = dpred.cpu().numpy()
    print("%-20s total %.8g vs %.8g | dpred rel L2 %.3e max-rel %.3e | colsum rel %.3e" % (
        name, losses["total_loss"], float(L["total_loss"]), np.linalg.norm(d - gp) / np.linalg.norm(gp), np.abs(d - gp).max() / np.abs(gp).max(),
        np.linalg.norm(d.sum((0, 1, 2)) - gp.sum((0, 1, 2))) / np.linalg.norm(gp.sum((0, 1, 2)))))
    for k in ("visible_loss_pred", "hidden_loss_pred", "ws_loss"):
        if k in L:
            print("      %-20s %.8g vs %.8g" % (k, losses.get(k, 0), float(L[k])))
