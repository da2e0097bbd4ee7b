#!/usr/bin/env python3
"""Experiment: one stream x batch 32 vs two HIP streams x batch 16 (kernels of the two half-batches overlap, so the
tail of one layer's grid runs under the body of the other half's layer)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepgraphpose_amd import engine, synthetic

H, W, NJ = 480, 640, 4
wts = synthetic.make_weights(50, NJ, False, seed=0)
frames = torch.from_numpy(synthetic.make_frames(32, H, W, NJ, seed=1)).cuda()
nets = [engine.DGPNet(50, NJ, H, W, max_batch=32) for _ in range(2)]
for n in nets:
    n.load_weights(wts)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def one(K):
    for _ in range(K):
        nets[0].infer(frames, check_range=False)

def two(K, parts):
    B = 32 // parts
    for _ in range(K):
        for p in range(parts):
            with torch.cuda.stream(streams[p % 2]):
                nets[p % 2].infer(frames[p * B:(p + 1) * B], check_range=False)

for name, fn in (("1 stream x 32", lambda k: one(k)), ("2 streams x 16", lambda k: two(k, 2)), ("2 streams x 2 x 8", lambda k: two(k, 4))):
    fn(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(20); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-20s %.3f ms / 32 frames  %.0f frames/s" % (name, dt / 20 * 1e3, 32 * 20 / dt), flush=True)

def alt(K):
    for i in range(K):
        with torch.cuda.stream(streams[i % 2]):
            nets[i % 2].infer(frames, check_range=False)
alt(4); torch.cuda.synchronize()
t0 = time.perf_counter(); alt(20); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%-20s %.3f ms / 32 frames  %.0f frames/s" % ("alternate 2 x 32", dt / 20 * 1e3, 32 * 20 / dt), flush=True)

# three engines / three streams
nets.append(engine.DGPNet(50, NJ, H, W, max_batch=32)); nets[2].load_weights(wts); streams.append(torch.cuda.Stream())
def alt3(K):
    for i in range(K):
        with torch.cuda.stream(streams[i % 3]):
            nets[i % 3].infer(frames, check_range=False)
alt3(6); torch.cuda.synchronize()
t0 = time.perf_counter(); alt3(21); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%-20s %.3f ms / 32 frames  %.0f frames/s" % ("alternate 3 x 32", dt / 21 * 1e3, 32 * 21 / dt), flush=True)
# sustained 2-stream: 300 steps
alt(10); torch.cuda.synchronize()
t0 = time.perf_counter(); alt(300); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%-20s %.3f ms / 32 frames  %.0f frames/s" % ("alternate 2 x 32, 300 steps", dt / 300 * 1e3, 32 * 300 / dt), flush=True)
t0 = time.perf_counter(); one(300); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%-20s %.3f ms / 32 frames  %.0f frames/s" % ("1 stream x 32, 300 steps", dt / 300 * 1e3, 32 * 300 / dt), flush=True)
