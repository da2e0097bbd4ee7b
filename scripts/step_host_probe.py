#!/usr/bin/env python3
"""Where the host spends a training step: wall time of each enqueue section of Trainer.step against the final wait."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
sys.argv = [sys.argv[0], "1", "0", sys.argv[1] if len(sys.argv) > 1 else "f16"]
import importlib.util
spec = importlib.util.spec_from_file_location("bt", os.path.join(os.path.dirname(__file__), "bench_train.py"))
bt = importlib.util.module_from_spec(spec); spec.loader.exec_module(bt)
tr = bt.tr
from deepgraphpose_amd import train as T, _lib
for _ in range(8): tr.step(bt.frames, bt.batch, bt.hy, bt.S0, bt.ws, bt.ws_max, 2000.0, 50.0)
torch.cuda.synchronize()
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0) + time.perf_counter() - t; return r
    return w
lib = tr.lib
class P:
    def __getattr__(self, n):
        f = getattr(lib, n); return timed(n, f)
tr.lib = P()
orig_fb = tr._forward_backward
tr._forward_backward = timed("_forward_backward(total)", orig_fb)
N = 50
t0 = time.perf_counter()
for _ in range(N): tr.step(bt.frames, bt.batch, bt.hy, bt.S0, bt.ws, bt.ws_max, 2000.0, 50.0)
tot = time.perf_counter() - t0
print("step %.3f ms" % (tot / N * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print("  %-40s %.3f ms" % (k, v / N * 1e3))
