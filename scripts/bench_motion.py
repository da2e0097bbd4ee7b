#!/usr/bin/env python3
"""Motion-energy kernel: device throughput vs the HBM roofline and vs the reference's numpy loop.
Algorithmic bytes = n_frames * frame_bytes (every frame read once); prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgraphpose_amd import engine
from oracle import dgp_oracle as O

T, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 480, 640
g = torch.Generator(device="cuda").manual_seed(0)
clip = torch.randint(0, 256, (T, H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
for _ in range(2):
    me = engine.motion_energy(clip)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
torch.cuda.synchronize(); e0.record()
for _ in range(n):
    me = engine.motion_energy(clip)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
nbytes = T * H * W * 3
sample = clip[:64].cpu().numpy()
t0 = time.perf_counter(); ref = O.motion_energy(sample); cpu_s = time.perf_counter() - t0
assert np.array_equal(ref, me[:64])
print(json.dumps({"kernel": "motion_energy", "frames": T, "ms": round(ms, 3), "frames_per_s": round(T / ms * 1e3),
                  "GB_per_s": round(nbytes / ms / 1e6, 1), "hbm_peak_GB_per_s": 8000, "frac": round(nbytes / ms / 1e6 / 8000, 3),
                  "includes": "memset + kernel + D2H of the sums + float64 division", "cpu_numpy_frames_per_s": round(64 / cpu_s),
                  "bit_exact_vs_oracle_on_64_frames": True}))
