"""Random conv layers on H2 tensors (every loader of the cell kernels: pointwise, per-tap, halo walk; 64- and 128-column tiles; stride 1 / 2 with
slim's conv2d_same padding; dilation; fp32 or H2 residual, also the stride-2 subsample of a block's last unit; fp32 or H2 output) against a float64
reference of the same quantised operands.  Usage: python scripts/fuzz_conv_h2.py [n] [seed]"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgraphpose_amd import engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(n):
    N, H, W = int(rng.integers(1, 5)), int(rng.integers(2, 60)), int(rng.integers(2, 90))
    k = int(rng.choice([1, 1, 3]))
    Cin = int(rng.choice([32, 64, 128, 256, 512, 1024])) if k == 1 else int(rng.choice([32, 64, 128, 256]))
    Cout = int(rng.choice([64, 128, 256, 512]))
    stride = int(rng.choice([1, 1, 2]))
    d = 1 if (k == 1 or stride == 2) else int(rng.choice([1, 1, 2]))
    if stride == 2:                                   # conv2d_same (slim resnet_utils): explicit padding, then VALID
        ke = k + (k - 1) * (d - 1); pt = (ke - 1) // 2; pe = (ke - 1) - pt
        Ho, Wo = (H + ke - 1 - ke) // 2 + 1, (W + ke - 1 - ke) // 2 + 1
    else:
        pt = pe = d * (k - 1) // 2; Ho, Wo = H, W
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    x = torch.relu(torch.randn((N, H, W, Cin), device="cuda", generator=g)) * float(rng.uniform(0.5, 4.0))
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    scale = (1 + 0.1 * rng.standard_normal(Cout)).astype(np.float32); bias = (0.1 * rng.standard_normal(Cout)).astype(np.float32)
    xe = engine.h2_exp_for(float(x.abs().max())); xh = engine.f32_to_h2(x, xe); xq = engine.h2_to_f32(xh, xe).double()
    xp = F.pad(xq.permute(0, 3, 1, 2), (pt, pe, pt, pe))
    ref = F.conv2d(xp, torch.from_numpy(w).double().cuda().permute(3, 2, 0, 1), stride=stride, dilation=d).permute(0, 2, 3, 1)
    assert ref.shape[1:3] == (Ho, Wo), (ref.shape, Ho, Wo)
    ref = ref * torch.from_numpy(scale).double().cuda() + torch.from_numpy(bias).double().cuda()
    res_kind = int(rng.integers(0, 4))                # 0 none, 1 fp32 same grid, 2 H2 same grid, 3 H2 on the 2x grid (subsampled [::2, ::2])
    res_t, rexp, rstride = None, 0, 0
    if res_kind:
        rs = 2 if res_kind == 3 else 1
        res = torch.randn((N, Ho * rs - (rs - 1) * int(rng.integers(0, 2)), Wo * rs - (rs - 1) * int(rng.integers(0, 2)), Cout), device="cuda", generator=g) * 2.0
        rstride = rs
        if res_kind >= 2:
            rexp = engine.h2_exp_for(float(res.abs().max())); res_t = engine.f32_to_h2(res, rexp); rq = engine.h2_to_f32(res_t, rexp).double()
        else:
            res_t = res; rq = res.double()
        ref = ref + rq[:, ::rs, ::rs][:, :Ho, :Wo]
    relu = bool(rng.integers(0, 2))
    if relu:
        ref = torch.relu(ref)
    y_h2 = bool(rng.integers(0, 4))
    ye = engine.h2_exp_for(float(ref.abs().max())) if y_h2 else 0
    desc = "N %d %3d x %3d  k %d s %d d %d  %4d -> %3d  res %d (%s) relu %d out %s" % (N, H, W, k, stride, d, Cin, Cout, res_kind,
           "-" if res_t is None else "x".join(str(v) for v in res_t.shape[1:3]), relu, "H2 " if y_h2 else "f32")
    try:
        y, yr = engine.conv2d_h2(xh, xe, w, stride=stride, rate=d, pad_t=pt, pad_l=pt, out_hw=(Ho, Wo), scale=scale, bias=bias, residual=res_t,
                                 res_stride=rstride, res_is_h2=res_kind >= 2, res_exp=rexp, relu=relu, y_is_h2=y_h2, y_exp=ye)
    except Exception as e:      # noqa: BLE001 -- a rejected configuration is reported, not fatal
        # the documented rejections (include/dgp_hip.h): fp32 output beyond 1x1 / stride 1, and an H2 residual with an fp32 output
        expected = (not y_h2) and (k != 1 or stride != 1 or res_kind >= 2)
        bad += not expected
        print(("rej " if expected else "REJ?") + desc + "  " + str(e)[-90:], flush=True)
        continue
    yo = engine.h2_to_f32(y, ye).double() if y_h2 else y.double()
    err = float((yo - ref).abs().max() / ref.abs().max())
    amax_ok = abs(float(yr.max()) - float(ref.abs().max())) <= 1e-4 * float(ref.abs().max()) if relu else True
    ok = err < 2e-5 and amax_ok
    bad += not ok
    print("%s %s  err %.2g" % ("ok " if ok else "BAD", desc, err), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
