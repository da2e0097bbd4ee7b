"""Random layers through the trainer's layer-level entry points -- weight gradient (fp32-MFMA tiles, the 16-bit-pipe tile, the LDS-DMA tile with
good and failed scale predictions) and data gradient (scale, ReLU gate as fp32 or H2, add modes) -- against float64.
Usage: python scripts/fuzz_backward_layers.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from deepgraphpose_amd import engine
from test_backward_layers_gpu import _same_pads, _im2col64, _dgrad_ref
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(n):
    N, H, W = int(rng.integers(1, 6)), int(rng.integers(3, 50)), int(rng.integers(3, 70))
    k = int(rng.choice([1, 1, 3]))
    Cin = int(rng.choice([32, 64, 128, 256, 512])); Cout = int(rng.choice([64, 128, 136, 256, 512]))
    stride = int(rng.choice([1, 1, 2])); rate = 1 if (k == 1 or stride == 2) else int(rng.choice([1, 2]))
    pad_t, Ho = _same_pads(H, k, stride, rate); pad_l, Wo = _same_pads(W, k, stride, rate)
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    x = torch.relu(torch.randn((N, H, W, Cin), generator=g, device="cuda")) * float(rng.uniform(0.3, 3.0))
    dy = torch.randn((N, Ho, Wo, Cout), generator=g, device="cuda") * float(10.0 ** rng.uniform(-5, -1))
    dy[torch.rand((N, Ho, Wo, Cout), generator=g, device="cuda") < 0.4] = 0.0
    desc = "N %d %2d x %2d  k %d s %d r %d  %3d -> %3d" % (N, H, W, k, stride, rate, Cin, Cout)
    cols = _im2col64(x.double(), k, stride, rate, pad_t, pad_l, Ho, Wo)
    ref = (cols.t() @ dy.double().reshape(-1, Cout)).reshape(k, k, Cin, Cout)
    cref = dy.double().reshape(-1, Cout).sum(0)
    errs = {}
    for name, fn in (("wgrad", lambda: engine.conv2d_wgrad(x, dy, k, stride, rate, pad_t, pad_l, ranged=True)),
                     ("wgrad-f32", lambda: engine.conv2d_wgrad(x, dy, k, stride, rate, pad_t, pad_l, ranged=False))):
        dw, cs = fn()
        errs[name] = max(float((dw.double() - ref).abs().max() / ref.abs().max()), float((cs.double() - cref).abs().max() / (cref.abs().max() + 1e-30)))
    if Cout >= 128 and Cin * k * k >= 128 and Cout % 8 == 0:
        ratio = [(1.0, 1.0), (0.4, 3.0), (2.0 ** -6, 1.0), (1.0, 2.0 ** 8), (0.0, 1.0)][int(rng.integers(0, 5))]
        dw, cs = engine.conv2d_wgrad_shadow(x, dy, k, stride, rate, pad_t, pad_l, prev_ratio=ratio)
        errs["dma %s" % (ratio,)] = float((dw.double() - ref).abs().max() / ref.abs().max())
    # data gradient
    w = torch.randn((k, k, Cin, Cout), generator=g, device="cuda") / float(np.sqrt(k * k * Cin))
    scale = 1.0 + 0.1 * torch.randn(Cout, generator=g, device="cuda")
    mask = torch.relu(torch.randn((N, H, W, Cin), generator=g, device="cuda"))
    add_mode = int(rng.choice([0, 1, -2])) if stride == 1 else int(rng.choice([0, 1]))
    add = None
    if add_mode == 1:
        add = torch.randn((N, H, W, Cin), generator=g, device="cuda") * 1e-3
    elif add_mode == -2:
        add = torch.randn((N, (H + 1) // 2, (W + 1) // 2, Cin), generator=g, device="cuda") * 1e-3
    mask_h2 = bool(rng.integers(0, 2))
    dref = _dgrad_ref(dy.double(), (w * scale).double(), H, W, stride, rate, pad_t, pad_l)
    if add_mode == 1:
        dref = dref + add.double()
    elif add_mode == -2:
        dref[:, ::2, ::2] += add.double()
    dref = torch.where(mask > 0, dref, torch.zeros_like(dref))
    try:
        dx = engine.conv2d_dgrad(dy, w, (H, W), stride, rate, pad_t, pad_l, scale=scale, mask=mask, dx_add=add, add_mode=add_mode if add is not None else 1,
                                 ranged=True, mask_h2=mask_h2)
        errs["dgrad add %d gate %s" % (add_mode, "h2" if mask_h2 else "f32")] = float((dx.double() - dref).abs().max() / dref.abs().max())
    except Exception as e:      # noqa: BLE001
        # (the documented constraint of dgp_conv2d_dgrad: Cout % 32 == 0 -- the ragged Cout = 136 is for the weight-gradient tiles)
        # ... and an H2 gate needs Cin >= 64 (the 32-column tile reads fp32 gates only)
        expected = bool(Cout % 32) or (mask_h2 and Cin < 64)
        errs[("dgrad rejected as documented" if expected else "dgrad REJECTED " + str(e)[-60:])] = 0.0 if expected else 1.0
    ok = all(v < 1e-5 for v in errs.values())
    bad += not ok
    print(("ok  " if ok else "BAD ") + desc + "  " + "  ".join("%s %.1e" % kv for kv in errs.items()), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
