"""A 2x2 'sub-pixel' conv of the 16-bit tier's heads on H1 tensors against float64 on the same fp16 operands (the merged heads' backward of the trainer)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgraphpose_amd import engine
N, H, W, Cin, Cout = 11, 30, 40, 64, 2048
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((N, H, W, Cin), device="cuda", generator=g) * 1e-4
w = (np.random.default_rng(0).standard_normal((2, 2, Cin, Cout)) * 0.05).astype(np.float32)
xe = engine.h2_exp_for(float(x.abs().max()))
xh = engine.f32_to_h1(x, xe)
xq = engine.h1_to_f32(xh, xe).double()
xp = torch.zeros((N, H + 1, W + 1, Cin), dtype=torch.float64, device="cuda"); xp[:, :H, :W] = xq
wq = torch.from_numpy(w).cuda()
mx = float(wq.abs().max()); import math
we = 14 - math.floor(math.log2(mx))
wq = (wq * 2.0 ** we).to(torch.float16).double() * 2.0 ** -we
ref = sum(xp[:, a:a + H, b:b + W] @ wq[a, b] for a in range(2) for b in range(2))
ye = engine.h2_exp_for(float(ref.abs().max()))
y, rng = engine.conv2d_h1(xh, xe, w, pad_t=0, pad_l=0, out_hw=(H, W), y_is_h1=True, y_exp=ye)
y = engine.h1_to_f32(y, ye)
print("max err", float((y.double() - ref).abs().max()), "ref max", float(ref.abs().max()), "y max", float(y.abs().max()))
