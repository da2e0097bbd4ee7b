"""Time one engine tier on the bench workload (ResNet-50, 640x480, 4 keypoints, batch 32): steps on one stream, then with two batches in
flight, and the per-launch table of one instrumented step.  `python scripts/bench_tier.py f16|parity [--steps 100] [--table out.tsv]`."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepgraphpose_amd import engine                                    # noqa: E402
from deepgraphpose_amd.synthetic import make_frames, make_weights       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("tier", choices=["f16", "parity"])
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--hw", type=int, nargs=2, default=[480, 640])
ap.add_argument("--depth", type=int, default=50)
ap.add_argument("--nj", type=int, default=4)
ap.add_argument("--table", type=str, default="")
ap.add_argument("--offset-ms", type=float, default=0.0, help="two streams: hold the second engine's stream back once by this long (do the two batches in flight run the same layers at the same time?)")
ap.add_argument("--timing-only", action="store_true", help="timing-only ablation builds: the results are garbage, skip the range check")
args = ap.parse_args()
H, W = args.hw
B, NJ = args.batch, args.nj
wts = make_weights(args.depth, NJ, False, seed=0, head_std=0.05)
frames = torch.from_numpy(make_frames(B, H, W, NJ, seed=100)).cuda()
net = engine.DGPNet(args.depth, NJ, H, W, max_batch=B, tier=args.tier)
net.load_weights(wts)
out = torch.zeros((B, NJ, 5), device="cuda")
for _ in range(5):
    net.infer_packed(frames, out)
torch.cuda.synchronize()
assert args.timing_only or not net.range_status()[0]
t0 = time.perf_counter()
for _ in range(args.steps):
    net.infer_packed(frames, out)
torch.cuda.synchronize()
t1 = time.perf_counter()
ms1 = (t1 - t0) / args.steps * 1e3
print("tier %s one stream : %.3f ms per %d-frame step, %.0f frames/s" % (args.tier, ms1, B, B / ms1 * 1e3))
pipe = engine.DGPPipeline(args.depth, NJ, H, W, max_batch=B, n_streams=2, first=net, tier=args.tier)
pipe.nets[1].load_weights(wts)
outs = [torch.zeros((B, NJ, 5), device="cuda") for _ in range(2)]
for i in range(6):
    pipe.submit(frames, outs[i & 1])
pipe.join()
torch.cuda.synchronize()
t0 = time.perf_counter()
if args.offset_ms > 0:
    with torch.cuda.stream(pipe.streams[1]):
        torch.cuda._sleep(int(args.offset_ms * 1e-3 * 100e6))      # (ROCm: wall-clock ticks of 100 MHz)
for i in range(args.steps):
    pipe.submit(frames, outs[i & 1])
pipe.join()
torch.cuda.synchronize()
t1 = time.perf_counter()
ms2 = ((t1 - t0) * 1e3 - args.offset_ms) / args.steps
print("tier %s two streams: %.3f ms per step, %.0f frames/s" % (args.tier, ms2, B / ms2 * 1e3))
net.profile_begin(3)
for _ in range(3):
    net.infer_packed(frames, out)
torch.cuda.synchronize()
ns, table = net.profile_end()
tot = sum(ms for _, _, ms in table)
print("instrumented step: %.3f ms over %d launches" % (tot, len(table)))
rows = ["%-110s %8.1f GFLOP %8.1f us %7.1f TFLOP/s" % (n, fl / 1e9, ms * 1e3, fl / ms / 1e9 if ms > 0 else 0.0) for n, fl, ms in table]
print("\n".join(rows))
if args.table:
    with open(args.table, "w") as f:
        f.write("name\tgflop\tus\n" + "\n".join("%s\t%.3f\t%.2f" % (n, fl / 1e9, ms * 1e3) for n, fl, ms in table) + "\n")
