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


# ---- the chain / unit kernels on H1 tensors (dgp_chain_h1 / dgp_unit_h1: the 16-bit tier's instances of csrc/dgp_chain.hip) ----------------
# Reference: float64 on exactly the fp16 operands each stage multiplies -- activations as the cells hold them, weights rounded to fp16 on
# their panel's power-of-two scale, and every intermediate (R2, X') rounded to the H1 cell the kernel builds its next operand from.  What
# is left is fp32 accumulation, one fp16 rounding per stored value, and -- behind an intermediate -- the rare element whose fp16 rounding
# falls the other way (an ulp of one operand of a 64..512-term sum): per-element bound for the first stage, a max-relative bound behind it.

def _q16(t, e):
    """float64 tensor -> the value its H1 cell holds (fp16(x 2^e) 2^-e)"""
    return (t * 2.0 ** e).to(torch.float16).double() * 2.0 ** -e


def _wq16(w):
    e = _w_exp(w)
    return (torch.from_numpy(np.asarray(w)).cuda() * 2.0 ** e).to(torch.float16).double() * 2.0 ** -e


H1_CHAIN_CASES = [
    # N, Ho, Wo, C, C1, CIN2, res_mode
    (2, 17, 23, 64, 64, 0, 1),        # identity unit of block1 (ragged last tile)
    (3, 30, 40, 64, 64, 64, 0),       # block1 unit_1: conv3 + shortcut conv K-concatenated, then unit_2's conv1
    (2, 15, 20, 64, 128, 0, 2),       # stride-2 unit at the end of block1 -> block2 unit_1's conv1
    (2, 15, 20, 128, 128, 0, 1),      # identity unit of block2
    (2, 8, 10, 128, 256, 0, 2),       # end of block2 -> block3 unit_1's conv1
    (32, 60, 80, 128, 128, 0, 1),     # batch-32 640x480 shape of block2 (1200 tiles over persistent workgroups)
]


@pytest.mark.parametrize("case", H1_CHAIN_CASES)
def test_chain_on_h1_tensors(lib_built, case):
    from deepgraphpose_amd import engine
    N, Ho, Wo, C, C1, CIN2, res_mode = case
    seed = abs(hash(case)) % (2 ** 31)
    g = torch.Generator(device="cuda").manual_seed(seed)
    rng = np.random.default_rng(seed)
    C4, M = 4 * C, N * Ho * Wo
    r2 = torch.relu(torch.randn((N, Ho, Wo, C), device="cuda", generator=g)) * 3.0
    shp = (N, Ho, Wo, CIN2) if res_mode == 0 else ((N, Ho, Wo, C4) if res_mode == 1 else (N, 2 * Ho - 1, 2 * Wo, C4))
    src2 = torch.relu(torch.randn(shp, device="cuda", generator=g)) * 2.0
    w3 = (rng.standard_normal((C + CIN2, C4)) / np.sqrt(C + CIN2)).astype(np.float32)
    w1 = (rng.standard_normal((C4, C1)) / np.sqrt(C4)).astype(np.float32)
    s3 = None if res_mode == 0 else (1 + 0.1 * rng.standard_normal(C4)).astype(np.float32)
    b3 = (0.1 * rng.standard_normal(C4)).astype(np.float32)
    s1 = (1 + 0.1 * rng.standard_normal(C1)).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(C1)).astype(np.float32)
    dd = lambda a: torch.from_numpy(np.asarray(a)).double().cuda()
    if res_mode == 0:
        e_r2 = e_s2 = engine.h2_exp_for(max(float(r2.max()), float(src2.max())))
    else:
        e_r2, e_s2 = engine.h2_exp_for(float(r2.max())), engine.h2_exp_for(float(src2.max()))
    r2h, s2h = engine.f32_to_h1(r2, e_r2), engine.f32_to_h1(src2, e_s2)
    r2q, s2q = engine.h1_to_f32(r2h, e_r2).double(), engine.h1_to_f32(s2h, e_s2).double()
    if res_mode == 0:
        acc = torch.cat([r2q.reshape(M, C), s2q.reshape(M, CIN2)], 1) @ _wq16(w3) + dd(b3)
    else:
        sc = s2q if res_mode == 1 else s2q[:, ::2, ::2]
        acc = (r2q.reshape(M, C) @ _wq16(w3)) * dd(s3) + dd(b3) + sc.reshape(M, C4)
    x_ref = torch.relu(acc)
    e_x = engine.h2_exp_for(float(x_ref.max()))
    r1_ref = torch.relu((_q16(x_ref, e_x) @ _wq16(w1)) * dd(s1) + dd(b1))
    e_r1 = engine.h2_exp_for(float(r1_ref.max()))
    xo, r1, xrng, r1rng = engine.chain_h2(r2h, e_r2, s2h, e_s2, w3, s3, b3, w1, s1, b1, res_mode, e_x, e_r1, h1=True)
    assert xo.dtype == torch.float16 and r1.dtype == torch.float16
    x_out = engine.h1_to_f32(xo, e_x).double().reshape(M, C4)
    r1_out = engine.h1_to_f32(r1, e_r1).double().reshape(M, C1)
    mx, mr = float(x_ref.max()), float(r1_ref.max())
    tol = 2.0 ** -11 * x_ref.abs() * (1 + 1e-3) + 2.0 ** -24 * 2.0 ** -e_x + 4e-6 * mx
    assert bool(((x_out - x_ref).abs() <= tol).all()), (case, float(((x_out - x_ref).abs() - tol).max()))
    assert float((r1_out - r1_ref).abs().max()) < 3e-3 * mr, case                      # (an X' cell rounded the other way moves a sum by <= 2^-11 |x w|)
    assert float((r1_out - r1_ref).abs().mean()) < 2e-4 * mr, case
    assert abs(float(xrng.max()) - mx) <= 1e-4 * mx and abs(float(r1rng.max()) - mr) <= 2e-3 * mr
    if M > 100000 or res_mode == 0:
        return
    # the layer-by-layer H1 kernels on the same cells: same operands, another accumulation order
    y3, _ = engine.conv2d_h1(r2h, e_r2, w3.reshape(1, 1, C, C4), scale=s3, bias=b3, residual=s2h, res_stride=res_mode, res_exp=e_s2, relu=True,
                             y_is_h1=True, y_exp=e_x, out_hw=(Ho, Wo))
    y1, _ = engine.conv2d_h1(y3, e_x, w1.reshape(1, 1, C4, C1), scale=s1, bias=b1, relu=True, y_is_h1=True, y_exp=e_r1)
    a = engine.h1_to_f32(y3, e_x).double().reshape(M, C4)
    b = engine.h1_to_f32(y1, e_r1).double().reshape(M, C1)
    assert float(((a - x_out).abs() > 0).double().mean()) < 2e-3, case                 # all but the rare tie-side flips are bit-identical
    assert float((a - x_out).abs().max()) <= 2.0 ** -10 * mx and float((b - r1_out).abs().max()) < 3e-3 * mr, case


H1_UNIT_CASES = [
    # N, H, W, CIN2, res_mode            (C = C1 = 64: block1)
    (2, 13, 21, 0, 1),        # ragged in both directions: halo zeros at every image edge, the 8-pixel DMA groups past the tile
    (1, 4, 16, 0, 1),         # exactly one tile
    (2, 19, 33, 64, 0),       # unit_1: conv3 + shortcut conv K-concatenated
    (32, 120, 160, 0, 1),     # the batch-32 640x480 shape of block1/unit_2
    (4, 120, 160, 64, 0),     # block1/unit_1 at full frame size
]


@pytest.mark.parametrize("case", H1_UNIT_CASES)
def test_unit_kernel_on_h1_tensors(lib_built, case):
    from deepgraphpose_amd import engine
    N, H, W, CIN2, res_mode = case
    C = C1 = 64
    C4, M = 4 * C, N * H * W
    seed = abs(hash(case)) % (2 ** 31)
    g = torch.Generator(device="cuda").manual_seed(seed)
    rng = np.random.default_rng(seed)
    r1 = torch.relu(torch.randn((N, H, W, C), device="cuda", generator=g)) * 3.0
    src2 = torch.relu(torch.randn((N, H, W, CIN2 if res_mode == 0 else C4), device="cuda", generator=g)) * 2.0
    w2 = (rng.standard_normal((3, 3, C, C)) / np.sqrt(9 * C)).astype(np.float32)
    s2 = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    b2 = (0.1 * rng.standard_normal(C)).astype(np.float32)
    w3 = (rng.standard_normal((C + CIN2, C4)) / np.sqrt(C + CIN2)).astype(np.float32)
    s3 = None if res_mode == 0 else (1 + 0.1 * rng.standard_normal(C4)).astype(np.float32)
    b3 = (0.1 * rng.standard_normal(C4)).astype(np.float32)
    w1 = (rng.standard_normal((C4, C1)) / np.sqrt(C4)).astype(np.float32)
    s1 = (1 + 0.1 * rng.standard_normal(C1)).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(C1)).astype(np.float32)
    dd = lambda a: torch.from_numpy(np.asarray(a)).double().cuda()
    e_r1 = engine.h2_exp_for(float(r1.max()))
    r1h = engine.f32_to_h1(r1, e_r1)
    r1q = engine.h1_to_f32(r1h, e_r1).double()
    xp = torch.zeros((N, H + 2, W + 2, C), dtype=torch.float64, device="cuda")
    xp[:, 1:H + 1, 1:W + 1] = r1q
    r2_ref = torch.zeros((M, C), dtype=torch.float64, device="cuda")
    w2q = _wq16(w2).reshape(9, C, C)
    for t in range(9):                                    # tap by tap: the batch-32 case would need 2.8 GB of im2col columns in float64
        r2_ref += xp[:, t // 3:t // 3 + H, t % 3:t % 3 + W].reshape(M, C) @ w2q[t]
    r2_ref = torch.relu(r2_ref * dd(s2) + dd(b2))
    if res_mode == 0:
        e_r2 = e_s2 = engine.h2_exp_for(max(float(r2_ref.max()), float(src2.max())))
    else:
        e_r2, e_s2 = engine.h2_exp_for(float(r2_ref.max())), engine.h2_exp_for(float(src2.max()))
    s2h = engine.f32_to_h1(src2, e_s2)
    s2q = engine.h1_to_f32(s2h, e_s2).double()
    r2q = _q16(r2_ref, e_r2)
    if res_mode == 0:
        acc = torch.cat([r2q, s2q.reshape(M, CIN2)], 1) @ _wq16(w3) + dd(b3)
    else:
        acc = (r2q @ _wq16(w3)) * dd(s3) + dd(b3) + s2q.reshape(M, C4)
    x_ref = torch.relu(acc)
    e_x = engine.h2_exp_for(float(x_ref.max()))
    r1o_ref = torch.relu((_q16(x_ref, e_x) @ _wq16(w1)) * dd(s1) + dd(b1))
    e_o = engine.h2_exp_for(float(r1o_ref.max()))
    xo, r1o, r2rng, xrng, orng = engine.unit_h2(r1h, e_r1, s2h, e_s2, w2, s2, b2, e_r2, w3, s3, b3, w1, s1, b1, res_mode, e_x, e_o, h1=True)
    x_out = engine.h1_to_f32(xo, e_x).double().reshape(M, C4)
    r1_out = engine.h1_to_f32(r1o, e_o).double().reshape(M, C1)
    mx, mo = float(x_ref.max()), float(r1o_ref.max())
    # X' sits behind the rounded R2, R1' behind the rounded X': max-relative bounds (see the note above), tight mean bounds
    assert float((x_out - x_ref).abs().max()) < 3e-3 * mx and float((x_out - x_ref).abs().mean()) < 2e-4 * mx, case
    assert float((r1_out - r1o_ref).abs().max()) < 3e-3 * mo and float((r1_out - r1o_ref).abs().mean()) < 2e-4 * mo, case
    for got, want in ((r2rng, r2_ref), (xrng, x_ref), (orng, r1o_ref)):
        assert abs(float(got.max()) - float(want.max())) <= 2e-3 * float(want.max())


def test_tier_f16_network_is_the_same_with_and_without_its_chain_kernels(lib_built, tmp_path):
    """DGP_CHAIN_H1 is read once per process: the layer-by-layer 16-bit engine runs in a child process.  Same fp16 operands, another
    accumulation order and intermediate cells rounded the other way -- which makes the two runs two realisations of the tier's own rounding
    noise (measured 0.027 px apart; each is 0.02-0.035 px from the oracle): the bound is the tier's band, not a bit-identity claim."""
    import os, subprocess, sys
    from deepgraphpose_amd import engine
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np, torch\n"
            "from deepgraphpose_amd import engine\n"
            "from deepgraphpose_amd.synthetic import make_frames, make_weights\n"
            "net = engine.DGPNet(50, 4, 192, 256, max_batch=3, tier='f16')\n"
            "net.load_weights(make_weights(50, 4, False, seed=5, head_std=0.05))\n"
            "mu, conf, idx = net.infer(torch.from_numpy(make_frames(3, 192, 256, 4, seed=6)).cuda(), 1.0, 1)\n"
            "np.savez(sys.argv[1], mu=mu.cpu().numpy(), conf=conf.cpu().numpy(), idx=idx.cpu().numpy())\n")
    out = {}
    for flag in ("0", "1"):
        path = str(tmp_path / ("t%s.npz" % flag))
        subprocess.check_call([sys.executable, "-c", code, path], env=dict(os.environ, DGP_CHAIN_H1=flag, PYTHONPATH=root), cwd=root)
        out[flag] = np.load(path)
    d = np.abs(out["0"]["mu"] - out["1"]["mu"]).max() * 8.0
    assert d < 0.08, d
    assert np.abs(out["0"]["conf"] - out["1"]["conf"]).max() < 0.02
    assert (out["0"]["idx"] == out["1"]["idx"]).all(-1).mean() > 0.9


def test_conv_cases_on_the_256_row_tile(lib_built):
    """Round 6: the 256 x 128 tile (four compute waves of 64 x 128).  DGP_W64 is read once per process, so the layer cases above run
    again in a child process with DGP_W64=2, which routes every H1 -> H1 convolution the tile can take to it (pointwise, strided,
    dilated, with same-grid and subsampled residuals, ragged last tiles)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider", "-k",
                        "conv_on_h1_tensors_matches_float64"], env=dict(os.environ, DGP_W64="2", PYTHONPATH=root), cwd=root,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


def test_tier_f16_network_is_bit_identical_on_the_256_row_tile(lib_built, tmp_path):
    """The 256-row tile multiplies the same fp16 cells in the same order as the 128-row tile (K-steps in order, odd k-groups before even
    ones inside a step, one accumulator per output): the network's outputs do not change by a bit whichever tile a layer runs on.  The
    child process also checks that the tile really ran (the profile table names the kernel of every launch)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, os, numpy as np, torch\n"
            "from deepgraphpose_amd import engine\n"
            "from deepgraphpose_amd.synthetic import make_frames, make_weights\n"
            "net = engine.DGPNet(50, 4, 256, 320, max_batch=4, tier='f16')\n"
            "net.load_weights(make_weights(50, 4, False, seed=5, head_std=0.05))\n"
            "fr = torch.from_numpy(make_frames(4, 256, 320, 4, seed=6)).cuda()\n"
            "mu, conf, idx = net.infer(fr, 1.0, 1)\n"
            "net.profile_begin(1); net.infer(fr, 1.0, 1); torch.cuda.synchronize(); ns, table = net.profile_end()\n"
            "n256 = sum('h1_256x128_k64' in n for n, _, _ in table)\n"
            "assert (n256 > 0) == (os.environ['DGP_W64'] == '2'), (n256, [n for n, _, _ in table])\n"
            "np.savez(sys.argv[1], mu=mu.cpu().numpy(), conf=conf.cpu().numpy(), idx=idx.cpu().numpy())\n")
    out = {}
    for flag in ("0", "2"):
        path = str(tmp_path / ("w%s.npz" % flag))
        subprocess.check_call([sys.executable, "-c", code, path], env=dict(os.environ, DGP_W64=flag, PYTHONPATH=root), cwd=root)
        out[flag] = np.load(path)
    for k in ("mu", "conf", "idx"):
        assert np.array_equal(out["0"][k], out["2"][k]), k
