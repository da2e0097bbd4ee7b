"""Random conv layers on fp32 activations (dgp_conv2d and dgp_conv2d_ranged: the tiles of the round-1 path and of the trainer's forward -- small
Cin / Cout generic tiles, 64- and 128-column tiles, the fp16-split kernels when ranges are given), max-pool and the H2 converters, against float64.
Usage: python scripts/fuzz_conv_f32.py [n] [seed]"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgraphpose_amd import engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(n):
    N, H, W = int(rng.integers(1, 5)), int(rng.integers(1, 50)), int(rng.integers(1, 70))
    k = int(rng.choice([1, 1, 3, 3, 7]))
    Cin = int(rng.choice([4, 8, 16, 32, 64, 128, 256, 1024])) if k < 7 else 4
    Cout = int(rng.choice([4, 8, 12, 16, 32, 64, 96, 128, 256, 512]))
    stride = int(rng.choice([1, 1, 2]))
    d = 1 if (k != 3 or stride == 2) else int(rng.choice([1, 2, 4]))
    ke = k + (k - 1) * (d - 1)
    if stride == 2:
        pt = (ke - 1) // 2; pe = (ke - 1) - pt
    else:
        pt = pe = (ke - 1) // 2
    Ho, Wo = (H + pt + pe - ke) // stride + 1, (W + pt + pe - ke) // stride + 1
    if Ho < 1 or Wo < 1:
        continue
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn((N, H, W, Cin), device="cuda", generator=g) * float(10.0 ** rng.uniform(-2, 2))
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    scale = (1 + 0.1 * rng.standard_normal(Cout)).astype(np.float32) if rng.integers(0, 2) else None
    bias = (0.1 * rng.standard_normal(Cout)).astype(np.float32) if rng.integers(0, 2) else None
    res_kind = int(rng.integers(0, 3))          # 0 none, 1 same grid, 2 the 2x grid (subsampled)
    xp = F.pad(x.double().permute(0, 3, 1, 2), (pt, pe, pt, pe))
    ref = F.conv2d(xp, torch.from_numpy(w).double().cuda().permute(3, 2, 0, 1), stride=stride, dilation=d).permute(0, 2, 3, 1)
    if scale is not None: ref = ref * torch.from_numpy(scale).double().cuda()
    if bias is not None: ref = ref + torch.from_numpy(bias).double().cuda()
    res_t, rstride = None, 0
    pre = ref.abs()                                   # |conv * scale + bias| before the residual: the fp32 addition rounds relative to its operands
    if res_kind:
        rs = res_kind
        res_t = torch.randn((N, Ho * rs - (rs - 1) * int(rng.integers(0, 2)), Wo * rs - (rs - 1) * int(rng.integers(0, 2)), Cout), device="cuda", generator=g) * float(ref.abs().max()) * 0.3
        rstride = rs
        ref = ref + res_t.double()[:, ::rs, ::rs][:, :Ho, :Wo]
        pre = pre + res_t.double().abs()[:, ::rs, ::rs][:, :Ho, :Wo]
    relu = bool(rng.integers(0, 2))
    if relu: ref = torch.relu(ref)
    # error bound per element: what fp32 operands allow, sum |x w| * 2^-21 (+ the residual's rounding)
    absx = F.pad(x.double().abs().permute(0, 3, 1, 2), (pt, pe, pt, pe))
    bound = F.conv2d(absx, torch.from_numpy(np.abs(w)).double().cuda().permute(3, 2, 0, 1), stride=stride, dilation=d).permute(0, 2, 3, 1)
    if scale is not None: bound = bound * torch.from_numpy(np.abs(scale)).double().cuda()
    bound = bound * 2.0 ** -20 + pre * 2.0 ** -22 + 1e-30
    desc = "N %d %2d x %2d  k %d s %d d %d  %4d -> %3d  res %d relu %d" % (N, H, W, k, stride, d, Cin, Cout, res_kind, relu)
    ok, msg = True, ""
    for ranged in (False, True):
        try:
            y = engine.conv2d(x, w, stride, d, pt, pt, (Ho, Wo), scale, bias, res_t, rstride, relu, ranged=ranged)
        except Exception as e:      # noqa: BLE001
            ok = False; msg += " %s: REJECTED %s" % ("ranged" if ranged else "plain", str(e)[-70:]); continue
        r = float(((y.double() - ref).abs() / bound).max())
        if not (r <= 1.0 and torch.isfinite(y).all()):
            ok = False
        msg += " %s %.2f" % ("ranged" if ranged else "plain", r)
    bad += not ok
    print(("ok  " if ok else "BAD ") + desc + "  err / bound:" + msg, flush=True)
    # max-pool and the H2 round trip on the same activation tensor
    if Cin % 8 == 0 and it % 3 == 0:
        mp = engine.maxpool_3x3s2_same(x)
        # SAME padding of a 3x3 / 2 pool: total pad = max((ceil(H / 2) - 1) * 2 + 3 - H, 0), the smaller half first; padded cells never win (-inf)
        ph = max(((H + 1) // 2 - 1) * 2 + 3 - H, 0); pw = max(((W + 1) // 2 - 1) * 2 + 3 - W, 0)
        mref = F.max_pool2d(F.pad(x.permute(0, 3, 1, 2), (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=float("-inf")), 3, 2).permute(0, 2, 3, 1)
        e = engine.h2_exp_for(float(x.abs().max()))
        back = engine.h2_to_f32(engine.f32_to_h2(x.contiguous(), e), e)
        rt = float(((back - x).abs() / x.abs().max()).max())
        ok2 = torch.equal(mp, mref.contiguous()) and rt < 2.0 ** -21
        bad += not ok2
        if not ok2:
            print("BAD  pool / H2 round trip on N %d %d x %d x %d: pool equal %s, round trip %.2e of the maximum" % (N, H, W, Cin, torch.equal(mp, mref.contiguous()), rt), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
