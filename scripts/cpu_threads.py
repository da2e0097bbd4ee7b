"""The CPU oracle's frames/s against torch's thread count (8 ... 128) on the bench workload: how many host threads bench.py's cpu_baseline leg should use."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from oracle import dgp_oracle as O
from deepgraphpose_amd.synthetic import make_weights, make_frames
w = make_weights(50, 4, False, 0); fr = make_frames(4, 480, 640)
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    O.infer(fr[:1], w)
    t = time.perf_counter(); O.infer(fr, w); dt = time.perf_counter() - t
    print(th, "threads:", 4 / dt, "fps", flush=True)
