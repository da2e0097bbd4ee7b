import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgraphpose_amd import engine
g = torch.Generator(device="cuda").manual_seed(1)
N, H, W, Cin, Cout = 1, 8, 16, 128, 128
dy = torch.randn((N, H, W, Cout), generator=g, device="cuda") * 1e-3
w = torch.randn((1, 1, Cin, Cout), generator=g, device="cuda") / float(np.sqrt(Cin))
mask = torch.relu(torch.randn((N, H, W, Cin), generator=g, device="cuda"))
a = engine.conv2d_dgrad(dy, w, (H, W), mask=mask, ranged=True)
b = engine.conv2d_dgrad(dy, w, (H, W), mask=mask, ranged=True, mask_h2=True)
c = engine.conv2d_dgrad(dy, w, (H, W), mask=None, ranged=True)
print("fp32 gate vs none-gated*mask:", float((a - c * (mask > 0)).abs().max()))
za, zb = (a == 0), (b == 0)
print("gate density fp32 %.4f h2 %.4f  mismatching gates %.4f" % (float(za.float().mean()), float(zb.float().mean()), float((za != zb).float().mean())))
A, B = za[0].cpu().numpy(), zb[0].cpu().numpy()
mm = (A != B)
print("mismatch by channel mod 8:", [round(float(mm[..., k::8].mean()), 3) for k in range(8)])
print("mismatch by pixel (first 32):", [round(float(v), 2) for v in mm.reshape(-1, Cin).mean(1)[:32]])
# is B the gate of some other position?
M = (mask[0] > 0).cpu().numpy()
for shift in range(-8, 9):
    r = np.roll(M, shift, axis=-1)
    print("channel shift", shift, "agreement of h2 gate with rolled true gate: %.3f" % float(((~B) == r).mean()))
