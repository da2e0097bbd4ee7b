#!/usr/bin/env python3
"""BASELINE configs[4] shape check: ResNet-101, 1280x720 frames, 20 keypoints, one GPU.  Prints frames/s and parity of
a few frames against the CPU oracle (slow: the oracle takes ~10 s per frame at this size).
Usage: bench_r101.py [batch] [--json] [--parity] [--tier f16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgraphpose_amd import engine, synthetic
from deepgraphpose_amd.arch import conv_macs_per_frame

H, W, NJ, B = 720, 1280, 20, int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
TIER = sys.argv[sys.argv.index("--tier") + 1] if "--tier" in sys.argv else "parity"
PEAK = 2500.0 if TIER == "f16" else 2500.0 / 3.0
wts = synthetic.make_weights(101, NJ, False, seed=0)
frames = synthetic.make_frames(B, H, W, NJ, seed=1)
net = engine.DGPNet(101, NJ, H, W, max_batch=B, tier=TIER)
net.load_weights(wts)
f = torch.from_numpy(frames).cuda()
for _ in range(2):
    mu, conf, idx = net.infer(f, check_range=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    mu, conf, idx = net.infer(f, check_range=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
gf = 2.0 * conv_macs_per_frame(H, W, 101, NJ, False) / 1e9
print("ResNet-101 %dx%d nj=%d batch %d: %.2f ms/step, %.1f frames/s, %.1f TFLOP/s of algorithmic conv FLOPs (%.1f GFLOP/frame)"
      % (W, H, NJ, B, dt * 1e3, B / dt, gf * B / dt / 1e3, gf), flush=True)
if "--json" in sys.argv:
    import json
    # the headline's arrangement as well: two engines on two HIP streams, batches dealt in turn (engine.DGPPipeline)
    pipe = engine.DGPPipeline(101, NJ, H, W, max_batch=B, n_streams=2, first=net, tier=TIER)
    pipe.nets[1].load_weights(wts)
    pipe.calibrate(f)
    outs = [torch.zeros((B, NJ, 5), dtype=torch.float32, device="cuda") for _ in range(2)]
    for i in range(4):
        pipe.submit(f, outs[i & 1], 1.0, 1)
    pipe.join(); torch.cuda.synchronize()
    K2 = 10
    t0 = time.perf_counter()
    for i in range(K2):
        pipe.submit(f, outs[i & 1], 1.0, 1)
    pipe.join(); torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / K2
    assert not pipe.range_status()[0]
    print(json.dumps({"frames_per_s": round(B / dt2, 1), "ms_per_step": round(dt2 * 1e3, 3), "batch": B, "steps": K2, "streams": 2,
                      "conv_tflops": round(gf * B / dt2 / 1e3, 1), "frac_of_its_peak": round(gf * B / dt2 / 1e3 / PEAK, 4), "tier": TIER,
                      "one_stream": {"frames_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3), "steps": K},
                      "algorithmic_gflop_per_frame": round(gf, 1),
                      "workload": "BASELINE configs[4], per-GPU shape: ResNet-101, 1280x720, 20 keypoints, batch %d, two batches in flight on two HIP "
                                  "streams like the headline number (one_stream: a single engine)" % B}), flush=True)
if "--parity" in sys.argv:
    from oracle import dgp_oracle as O
    ref = O.infer(frames[:1], wts, 101, 8.0, 1.0, 1)
    d = np.abs(mu[:1].cpu().numpy() - ref["mu"]).max() * 8.0
    print("parity on 1 frame vs oracle: max |d| = %.2e px, idx exact: %s" % (d, np.array_equal(idx[:1].cpu().numpy(), ref["idx"])))
