#!/usr/bin/env python3
"""Probe: does the PHASE between the two engines' streams matter?  Two engines, batches dealt in turn (as engine.DGPPipeline does); stream 1 is
delayed once by a spin kernel of `d` GPU cycles before the timed run, so that its layer sequence runs `d` behind stream 0's.
python scripts/stream_offset_probe.py [parity|f16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepgraphpose_amd import engine, synthetic
tier = sys.argv[1] if len(sys.argv) > 1 else "parity"
H, W, NJ, B = 480, 640, 4, 32
wts = synthetic.make_weights(50, NJ, False, seed=0, head_std=0.05)
frames = torch.from_numpy(synthetic.make_frames(B, H, W, NJ, seed=100)).cuda()
nets = [engine.DGPNet(50, NJ, H, W, max_batch=B, tier=tier) for _ in range(2)]
outs = [torch.zeros((B, NJ, 5), device="cuda") for _ in range(2)]
for n in nets:
    n.load_weights(wts)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for i in range(2):
    with torch.cuda.stream(streams[i]):
        for _ in range(3):
            nets[i].infer_packed(frames, outs[i])
torch.cuda.synchronize()
def run(K, delay):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if delay:
        with torch.cuda.stream(streams[1]):
            torch.cuda._sleep(int(delay))
    for i in range(K):
        with torch.cuda.stream(streams[i % 2]):
            nets[i % 2].infer_packed(frames, outs[i % 2])
    torch.cuda.synchronize()
    return time.perf_counter() - t0
run(40, 0)
for rep in range(2):
    for d in (0, 1e6, 2e6, 3e6, 4e6, 5e6, 6e6, 8e6, 10e6, 12e6):
        K = 200
        dt = run(K, d)
        print("tier %s delay %5.1f Mcycles: %.3f ms per step, %.0f frames/s" % (tier, d / 1e6, dt / K * 1e3, B * K / dt), flush=True)
