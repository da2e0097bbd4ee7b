#!/usr/bin/env python3
"""Per-phase s_memtime shares of wgrad_h3 at the training step's shapes (needs the -DDGP_DIAG build: DGP_HIP_LIB=...).
Also times each shape with hipEvents on the normal build (no DGP_HIP_LIB) for reference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepgraphpose_amd import engine

# (name, N, H, W, Cin, Cout, k, stride, rate) -- ResNet-50 os16 at 480x640, 11 frames
SHAPES = [
    ("b1.conv2 3x3 64->64 (f32<1> tile)", 11, 120, 160, 64, 64, 3, 1, 1),
    ("b1.conv3 1x1 64->256", 11, 120, 160, 64, 256, 1, 1, 1),
    ("b2.conv2 3x3 128->128", 11, 60, 80, 128, 128, 3, 1, 1),
    ("b2.conv3 1x1 128->512", 11, 60, 80, 128, 512, 1, 1, 1),
    ("b3.conv1 1x1 1024->256", 11, 30, 40, 1024, 256, 1, 1, 1),
    ("b3.conv2 3x3 256->256", 11, 30, 40, 256, 256, 3, 1, 1),
    ("b4.conv2 3x3 512->512 rate 2", 11, 30, 40, 512, 512, 3, 1, 2),
    ("b4.conv3 1x1 512->2048", 11, 30, 40, 512, 2048, 1, 1, 1),
]
g = torch.Generator(device="cpu").manual_seed(0)
for name, N, H, W, Cin, Cout, k, st, rate in SHAPES:
    x = torch.randn(N, H, W, Cin, generator=g).cuda()
    dy = torch.randn(N, H, W, Cout, generator=g).cuda()
    pad = rate * (k // 2)
    print("==", name, flush=True)
    for _ in range(2):
        engine.conv2d_wgrad(x, dy, k, st, rate, pad, pad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        engine.conv2d_wgrad(x, dy, k, st, rate, pad, pad)
    e1.record(); torch.cuda.synchronize()
    fl = 2.0 * N * H * W * Cin * Cout * k * k
    ms = e0.elapsed_time(e1) / 5
    print("   %.3f ms per call incl. absmax x2 + memset (%.1f TFLOP/s)" % (ms, fl / ms / 1e9), flush=True)
