#!/usr/bin/env python3
"""Run one forward of the bench workload under the -DDGP_DIAG build and print the per-layer stamp summaries.
python scripts/diag_net.py [parity|f16]   (DGP_HIP_LIB=<diag build>; DGP_CHAIN=0 for the layer-by-layer parity engine)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepgraphpose_amd import engine, synthetic
net = engine.DGPNet(50, 4, 480, 640, max_batch=32, tier=(sys.argv[1] if len(sys.argv) > 1 else "parity"))
net.load_weights(synthetic.make_weights(50, 4, False, seed=0))
f = torch.from_numpy(synthetic.make_frames(32, 480, 640, 4, seed=1)).cuda()
for _ in range(2):
    net.infer(f, check_range=False)
torch.cuda.synchronize()
