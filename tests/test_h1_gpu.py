"""The 16-bit tier (include/dgp_hip.h, "H1"; dgp_net_set_tier(net, 1)): 2-byte activation cells -- fp16(x * 2^exp), plain NHWC fp16 with
the engine's calibrated per-tensor scales -- from the pool output to the block4 features, fp16 weight cells, ONE MFMA per product, fp32
accumulation / epilogues / heads / soft-argmax.  A REPORTED tier: its distance from the oracle is measured and pinned here, it is not
inside the 1e-3 px gate and nothing else in the suite runs on it.

Layer tests: the cell kernels on H1 tensors against a float64 reference that convolves exactly the fp16 operands the kernel multiplies
(activations as stored, weights rounded to fp16 on the panel's power-of-two scale): what is left is fp32 accumulation and the one fp16
rounding of the output cell."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _w_exp(w):
    """exponent e with max |w| 2^e in [2^14, 2^15): the weight cells' scale (pow2_scale_for)"""
    import math
    m = float(np.abs(w).max())
    return 14 - (math.frexp(m)[1] - 1)


def test_h1_cells_are_the_high_cells_of_h2(lib_built):
    from deepgraphpose_amd import engine
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((2, 9, 64), device="cuda", generator=g) * 11.0
    x[0, 0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 1e-6, 300.0, 123.456, -7.25], device="cuda")
    e = engine.h2_exp_for(float(x.abs().max()))
    h1 = engine.f32_to_h1(x, e)
    assert h1.dtype == torch.float16 and h1.shape == x.shape
    # plain NHWC fp16 of x * 2^e, round to nearest even
    assert torch.equal(h1, (x * 2.0 ** e).to(torch.float16))
    raw2 = engine.f32_to_h2(x, e).view(torch.int32).cpu().numpy().reshape(-1, 8)[:, :4].copy().view(np.float16)
    assert np.array_equal(raw2.reshape(-1), h1.cpu().numpy().reshape(-1))          # bit for bit the H2 pair's high cell
    back = engine.h1_to_f32(h1, e)
    assert float((back - x).abs().max()) <= 2.0 ** -11 * float(x.abs().max())


H1_CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, rate, residual (0 none, 2 same grid H1, 3 strided H1), y_is_h1
    (2, 17, 23, 64, 256, 1, 1, 1, 2, True),        # conv3 of an identity unit: 128 x 128 tile, pointwise loader, ONE K-step of 64 channels
    (1, 30, 40, 256, 64, 1, 1, 1, 0, True),        # conv1 -> 64 channels: the 128 x 64 tile
    (2, 19, 21, 64, 64, 3, 1, 1, 0, True),         # 3x3 on the 128 x 64 tile (per-tap loaders), one K-step per tap
    (2, 19, 21, 64, 64, 3, 2, 1, 0, True),         # strided 3x3
    (1, 15, 20, 128, 128, 3, 1, 2, 0, True),       # dilated 3x3: the halo walk on a ring of 64-channel pixels
    (2, 30, 40, 256, 256, 3, 1, 1, 0, True),       # block3 conv2 (halo walk)
    (1, 30, 40, 512, 512, 3, 1, 2, 2, True),       # block4 conv2 (+ an H1 residual)
    (2, 20, 24, 256, 512, 1, 2, 1, 0, True),       # strided shortcut conv (per-tap loaders, 1 tap)
    (2, 10, 12, 512, 1024, 1, 1, 1, 3, True),      # conv3 of a strided unit: residual subsampled from the 2x finer grid
    (1, 30, 40, 2048, 128, 1, 1, 1, 0, False),     # the head's pointwise GEMM: H1 in, fp32 out
    (3, 7, 9, 128, 128, 3, 1, 1, 0, True),         # frames smaller than a tile, ragged last tile
    (1, 9, 200, 128, 128, 3, 1, 2, 0, True),       # too wide for the ring -> per-tap loaders
    (32, 30, 40, 1024, 256, 1, 1, 1, 0, True),     # batch-32 block3 conv1
]


@pytest.mark.parametrize("case", H1_CONV_CASES)
def test_conv_on_h1_tensors_matches_float64_on_the_same_fp16_operands(lib_built, case):
    from deepgraphpose_amd import engine
    N, H, W, Cin, Cout, k, stride, rate, res_kind, y_h1 = case
    seed = abs(hash(case)) % (2 ** 31)
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.relu(torch.randn((N, H, W, Cin), device="cuda", generator=g)) * 3.0
    rngw = np.random.default_rng(seed)
    w = (rngw.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    scale = (1 + 0.1 * rngw.standard_normal(Cout)).astype(np.float32)
    bias = (0.1 * rngw.standard_normal(Cout)).astype(np.float32)
    keff = (k - 1) * rate + 1
    pad = (keff - 1) // 2
    Ho = (H + 2 * pad - keff) // stride + 1 if stride > 1 else H
    Wo = (W + 2 * pad - keff) // stride + 1 if stride > 1 else W
    if k == 1 and stride > 1:
        pad, Ho, Wo = 0, (H + stride - 1) // stride, (W + stride - 1) // stride
    x_exp = engine.h2_exp_for(float(x.abs().max()))
    xh = engine.f32_to_h1(x, x_exp)
    xq = engine.h1_to_f32(xh, x_exp).double()                                  # the activations as the cells hold them
    we = _w_exp(w)
    wq = (torch.from_numpy(w).cuda() * 2.0 ** we).to(torch.float16).double() * 2.0 ** -we      # the weights as the weight cells hold them
    xp = torch.zeros((N, H + 2 * pad + stride, W + 2 * pad + stride, Cin), dtype=torch.float64, device="cuda")
    xp[:, pad:pad + H, pad:pad + W] = xq
    cols = [xp[:, a * rate: a * rate + (Ho - 1) * stride + 1: stride, b * rate: b * rate + (Wo - 1) * stride + 1: stride]
            for a in range(k) for b in range(k)]
    ref = torch.stack(cols, 3).reshape(N * Ho * Wo, k * k * Cin) @ wq.reshape(-1, Cout)
    ref = ref.reshape(N, Ho, Wo, Cout) * torch.from_numpy(scale).double().cuda() + torch.from_numpy(bias).double().cuda()
    res_t, res_exp, res_stride = None, 0, 0
    if res_kind:
        shape = (N, Ho, Wo, Cout) if res_kind == 2 else (N, 2 * Ho - 1, 2 * Wo, Cout)
        res = torch.randn(shape, device="cuda", generator=g) * 2.0
        res_stride = 1 if res_kind == 2 else 2
        res_exp = engine.h2_exp_for(float(res.abs().max()))
        res_t = engine.f32_to_h1(res, res_exp)
        rq = engine.h1_to_f32(res_t, res_exp).double()
        ref = ref + (rq if res_stride == 1 else rq[:, ::2, ::2])
    ref = torch.relu(ref)
    y_exp = engine.h2_exp_for(float(ref.abs().max()))
    y, yrng = engine.conv2d_h1(xh, x_exp, w, stride=stride, rate=rate, pad_t=pad, pad_l=pad, out_hw=(Ho, Wo), scale=scale, bias=bias,
                               residual=res_t, res_stride=res_stride, res_exp=res_exp, relu=True, y_is_h1=y_h1, y_exp=y_exp)
    mx = float(ref.abs().max())
    if y_h1:
        out = engine.h1_to_f32(y, y_exp).double()
        # one fp16 rounding of the stored value (half an ulp of ITS binade, subnormal floor 2^-25 of the scaled range) + fp32 accumulation
        tol = 2.0 ** -11 * ref.abs() * (1 + 1e-3) + 2.0 ** -24 * 2.0 ** -y_exp + 4e-6 * mx
        assert bool(((out - ref).abs() <= tol).all()), (case, float(((out - ref).abs() - tol).max()))
    else:
        assert float((y.double() - ref).abs().max()) < 4e-6 * mx, case
    assert abs(float(yrng.max()) - mx) <= 1e-4 * mx                            # the tracked range is max |out| BEFORE the fp16 rounding


def _tier_errors(net, frames, ref):
    ft = torch.from_numpy(frames).cuda()
    sc = net.forward(ft)
    mu, conf, idx = net.infer(ft, 1.0, 1)
    torch.cuda.synchronize()
    d = (mu.cpu().numpy().astype(np.float64) - ref["mu"]) * 8.0
    px = np.sqrt((d ** 2).sum(-1))
    return dict(px_max=float(px.max()), px_rmse=float(np.sqrt((px ** 2).mean())),
                sc_rel=float(np.abs(sc.cpu().numpy() - ref["scmap"]).max() / np.abs(ref["scmap"]).max()),
                idx_agree=float((idx.cpu().numpy() == ref["idx"]).all(-1).mean()),
                conf_max=float(np.abs(conf.cpu().numpy() - ref["likelihoods"]).max()))


@pytest.mark.parametrize("shape", [(96, 128, 3, 4, False), (480, 640, 4, 4, False), (747, 832, 5, 1, True)])
def test_tier_f16_network_stays_within_its_measured_band(lib_built, shape):
    """The whole engine on the 16-bit tier against the fp32 oracle: not the parity gate (1e-3 px) but a pinned band around what the tier
    measures -- coordinates within 0.1 px, scoremap within 1 % of its range, window indices agreeing on >= 90 % of the (frame, joint)
    pairs -- and the SAME engine switched back to the parity tier is inside the 1e-3 px gate again (the tiers share nothing that
    could leak: scales are re-calibrated on every switch)."""
    from deepgraphpose_amd import engine
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    from oracle import dgp_oracle as O
    h, w, nj, B, locref = shape
    wts = make_weights(50, nj, locref, seed=5, head_std=0.05)
    frames = make_frames(B, h, w, nj, seed=6)
    ref = O.infer(frames, wts, 50, 8.0, 1.0, 1)
    net = engine.DGPNet(50, nj, h, w, max_batch=B, with_locref=locref, tier="f16")
    assert net.tier == "f16"
    net.load_weights(wts)
    e16 = _tier_errors(net, frames, ref)
    print("tier f16 %dx%d:" % (w, h), e16)
    assert e16["px_max"] < 0.1 and e16["px_rmse"] < 0.05 and e16["sc_rel"] < 1e-2 and e16["idx_agree"] >= 0.9 and e16["conf_max"] < 1e-2
    assert e16["px_max"] > 1e-6                      # (it IS another arithmetic: a zero here would mean the tier switch did nothing)
    again = _tier_errors(net, frames, ref)
    assert again == e16                              # deterministic: frozen scales, no atomics in the data path
    if locref:
        sc, lr = net.forward(torch.from_numpy(frames).cuda(), want_locref=True)
        _, lref = O.pose_heads(ref["features"], wts, True)
        assert float(np.abs(lr.cpu().numpy() - lref).max() / np.abs(lref).max()) < 1e-2
    net.set_tier("parity")
    e32 = _tier_errors(net, frames, ref)
    assert e32["px_max"] < 1e-3 and e32["idx_agree"] == 1.0 and e32["sc_rel"] < 1e-4
    feats16 = None
    net.set_tier("f16")
    _, feats16 = net.forward(torch.from_numpy(frames).cuda(), want_features=True)
    fr = ref["features"]
    assert float(np.abs(feats16.cpu().numpy() - fr).max() / np.abs(fr).max()) < 2e-2      # the H1 -> fp32 copy-out of the block4 features


def test_tier_f16_overflow_is_detected_and_recovered(lib_built):
    """The 16-bit tier shares the calibrate / range-check / re-calibrate contract of the H2 engine."""
    from deepgraphpose_amd.engine import DGPNet
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    from oracle import dgp_oracle as O
    nj = 3
    wts = make_weights(50, nj, False, seed=31, head_std=0.05)
    net = DGPNet(50, nj, 64, 96, max_batch=4, tier="f16")
    net.load_weights(wts)
    flat = np.zeros((2, 64, 96, 3), dtype=np.uint8)
    flat[..., 0], flat[..., 1], flat[..., 2] = 124, 117, 104             # ~ the mean pixel: tiny activations, tiny calibrated ranges
    frames = make_frames(2, 64, 96, nj, seed=31)
    net.infer(torch.from_numpy(flat).cuda(), check_range=False)
    assert net.range_status() == (False, 1)
    net.infer(torch.from_numpy(frames).cuda(), check_range=False)         # real frames outgrow those scales
    ov, _ = net.range_status()
    assert ov
    mu, _, idx = net.infer(torch.from_numpy(frames).cuda())               # re-calibrates on this batch
    ref = O.infer(frames, wts, 50, 8.0, 1.0, 1)
    assert float(np.abs(mu.cpu().numpy() - ref["mu"]).max()) * 8.0 < 0.1
