#!/usr/bin/env python3
"""Soak of engine.DGPPipeline: 3000 batches of 32 frames dealt to two engines; outputs stay bit-identical, device memory flat."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from deepgraphpose_amd import engine, synthetic
H, W, NJ = 480, 640, 4
pipe = engine.DGPPipeline(50, NJ, H, W, max_batch=32)
pipe.load_weights(synthetic.make_weights(50, NJ, False, seed=0, head_std=0.05))
fr = [torch.from_numpy(synthetic.make_frames(32, H, W, NJ, seed=s)).cuda() for s in (1, 2)]
out = [torch.zeros((32, NJ, 5), device="cuda") for _ in range(2)]
pipe.submit(fr[0], out[0]); pipe.submit(fr[1], out[1]); pipe.join(); torch.cuda.synchronize()
ref = [o.clone() for o in out]
m0 = torch.cuda.memory_allocated()
t0 = time.perf_counter()
N = 3000
for i in range(N):
    pipe.submit(fr[i % 2], out[i % 2])
    if i % 500 == 499:
        pipe.join(); torch.cuda.synchronize()
        assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), i
pipe.join(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("3000 batches: %.1f frames/s, memory delta %d bytes, overflow %s" % (N * 32 / dt, torch.cuda.memory_allocated() - m0, pipe.range_status()))
