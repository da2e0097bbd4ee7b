#!/usr/bin/env python3
"""One rank of a data-parallel Trainer.step loop at BASELINE configs[3]'s shape (several ranks on ONE GPU: DGP_DIST_BACKEND=gloo).
python3 scripts/dp_step_worker.py <steps> <tier>   (RANK / WORLD_SIZE / MASTER_* in the environment; scripts/timeline_dp.sh)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from deepgraphpose_amd import dist as ddist
rank, local, world = ddist.init_from_env()
import torch.distributed as dist
from deepgraphpose_amd.train import Trainer
from deepgraphpose_amd.loss import DGPHyper
from deepgraphpose_amd.synthetic import make_weights, make_frames
from test_train_gpu import _make_loss_case
steps, tier = int(sys.argv[1]), sys.argv[2]
nj, nt, hw = 4, 11, (480, 640)
rng = np.random.default_rng(44 + rank)
batch, _ = _make_loss_case(rng, nt, 60, 80, nj, 1, 0.0, 2)
S0 = np.zeros((3, nj))
for l in range(3):
    S0[l, l], S0[l, l + 1] = 1, -1
tr = Trainer(50, nj, hw[0], hw[1], max_frames=nt, tier=tier)
tr.load_weights(make_weights(50, nj, True, seed=4, head_std=0.05))
ft = torch.from_numpy(make_frames(nt, hw[0], hw[1], nj, seed=4 + rank)).cuda()
ws, ws_max = rng.uniform(5, 20, 3), rng.uniform(10, 40, 3)
hy = DGPHyper(gm2=1, gm3=3)
for _ in range(3):
    tr.step(ft, batch, hy, S0, ws, ws_max, 1000.0, 50.0)
torch.cuda.synchronize(); dist.barrier()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(ft, batch, hy, S0, ws, ws_max, 1000.0, 50.0)
torch.cuda.synchronize()
print("rank %d: %.3f ms per data-parallel step (world %d, backend %s, tier %s); gradient groups (floats): %s"
      % (rank, (time.perf_counter() - t0) / steps * 1e3, world, dist.get_backend(), tier, tr.grad_groups()), flush=True)
dist.barrier()
dist.destroy_process_group()
