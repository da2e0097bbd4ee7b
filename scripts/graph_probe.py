#!/usr/bin/env python3
"""Does replaying one inference step as a HIP graph (torch.cuda.CUDAGraph around DGPNet.infer) shorten the step?
The engine's launches go to torch's current stream through the C-ABI, so stream capture records them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgraphpose_amd import engine, synthetic
B = 32
net = engine.DGPNet(50, 4, 480, 640, max_batch=B, tier=(sys.argv[1] if len(sys.argv) > 1 else "parity"))
net.load_weights(synthetic.make_weights(50, 4, False, seed=0))
f = torch.from_numpy(synthetic.make_frames(B, 480, 640, 4, seed=1)).cuda()
for _ in range(3):
    out = net.infer(f, check_range=False)
torch.cuda.synchronize()
def timeit(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t_plain = timeit(lambda: net.infer(f, check_range=False))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): net.infer(f, check_range=False)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gout = net.infer(f, check_range=False)
torch.cuda.synchronize()
t_graph = timeit(g.replay)
ref = net.infer(f, check_range=False)
g.replay(); torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(ref, gout))
print("plain %.3f ms  graph %.3f ms  (%.1f vs %.1f frames/s)  identical outputs: %s" % (t_plain, t_graph, B / t_plain * 1e3, B / t_graph * 1e3, same))
