"""Random 3x3 / stride-1 H2 conv shapes (the halo walk and its fall-backs) against a float64 reference.  Usage: python scripts/fuzz_halo.py [n] [seed]"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgraphpose_amd import engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for k in range(n):
    N, H, W = int(rng.integers(1, 5)), int(rng.integers(2, 70)), int(rng.integers(2, 130))
    Cin, Cout = int(rng.choice([32, 64, 128, 256])), int(rng.choice([128, 256, 384]))
    d = int(rng.choice([1, 1, 2, 3]))
    res_kind = int(rng.integers(0, 3))
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    x = torch.relu(torch.randn((N, H, W, Cin), device="cuda", generator=g)) * 3.0
    w = (rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    scale = (1 + 0.1 * rng.standard_normal(Cout)).astype(np.float32); bias = (0.1 * rng.standard_normal(Cout)).astype(np.float32)
    xe = engine.h2_exp_for(float(x.abs().max())); xh = engine.f32_to_h2(x, xe); xq = engine.h2_to_f32(xh, xe).double()
    ref = F.conv2d(xq.permute(0, 3, 1, 2), torch.from_numpy(w).double().cuda().permute(3, 2, 0, 1), padding=d, dilation=d).permute(0, 2, 3, 1)
    ref = ref * torch.from_numpy(scale).double().cuda() + torch.from_numpy(bias).double().cuda()
    res_t, rexp = None, 0
    if res_kind:
        res = torch.randn((N, H, W, Cout), device="cuda", generator=g) * 2.0
        if res_kind == 2:
            rexp = engine.h2_exp_for(float(res.abs().max())); res_t = engine.f32_to_h2(res, rexp); ref = ref + engine.h2_to_f32(res_t, rexp).double()
        else:
            res_t = res; ref = ref + res.double()
    ref = torch.relu(ref)
    ye = engine.h2_exp_for(float(ref.abs().max()))
    y, yr = engine.conv2d_h2(xh, xe, w, stride=1, rate=d, pad_t=d, pad_l=d, out_hw=(H, W), scale=scale, bias=bias, residual=res_t,
                             res_stride=1 if res_kind else 0, res_is_h2=res_kind == 2, res_exp=rexp, relu=True, y_is_h2=True, y_exp=ye)
    err = float((engine.h2_to_f32(y, ye).double() - ref).abs().max() / ref.abs().max())
    ok = err < 2e-5
    bad += not ok
    print("%s N %d %3d x %3d  %3d -> %3d  dil %d res %d  strip %4d  err %.2g" % ("ok " if ok else "BAD", N, H, W, Cin, Cout, d, res_kind, 128 + 2 * d * (W + 1), err), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
