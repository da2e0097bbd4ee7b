"""GPU parity tests: HIP path (through the C-ABI) vs the CPU oracle on identical seeded inputs.

Tolerances (north_star): integer indices bit-exact; soft-argmax coordinates within 1e-3 px
(= 1.25e-4 scoremap cells at stride 8); layer outputs within fp32 accumulation-order noise.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PX_TOL = 1e-3          # px, north_star
STRIDE = 8.0


@pytest.fixture(scope="module")
def eng(lib_built):
    from deepgraphpose_amd import engine
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return engine


# settled A/B switches: compiled-in defaults in the product build, read from the environment only in a -DDGP_TUNING build (dgp_tune)
_TUNING_ONLY = {"DGP_STEM_ROWS", "DGP_PRESPLIT_WEIGHTS", "DGP_STEM_FUSED", "DGP_DMA", "DGP_COMPUTE_SPLIT"}


def _skip_unless_tuning_build(extra):
    if _TUNING_ONLY & set(extra):
        from deepgraphpose_amd import _lib
        if not _lib.load().dgp_tuning_build():
            pytest.skip("switch of -DDGP_TUNING builds: the product build compiles its default in")


def _rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


# ---------------------------------------------------------------------------- layers
CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, rate, same_explicit
    (2, 17, 23, 64, 64, 1, 1, 1),
    (2, 17, 23, 64, 256, 1, 1, 1),
    (1, 30, 40, 256, 64, 1, 1, 1),
    (2, 19, 21, 64, 64, 3, 1, 1),
    (2, 19, 21, 64, 64, 3, 2, 1),
    (1, 15, 20, 128, 128, 3, 1, 2),
    (1, 9, 11, 512, 512, 3, 1, 2),
    (2, 20, 24, 256, 512, 1, 2, 1),
    (1, 33, 47, 4, 64, 7, 2, 1),
    (1, 8, 8, 2048, 16, 2, 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_layer_matches_oracle(eng, case):
    from oracle import dgp_oracle as O
    N, H, W, Cin, Cout, k, stride, rate = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    scale = (1 + 0.1 * rng.standard_normal(Cout)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(Cout)).astype(np.float32)
    xt = O._to_nchw(x)
    if k == 2:      # the head's 2x2 conv: pad 1 before, 0 after
        ref = O.conv2d(torch.nn.functional.pad(xt, (1, 0, 1, 0)), w, 1, 1, "VALID")
        pad = 1
    else:
        ref = O.conv2d_same(xt, w, stride, rate)
        keff = (k - 1) * rate + 1
        pad = (keff - 1) // 2 if stride > 1 else O.tf_same_pads(H, k, stride, rate)[1]
    ref = O._to_nhwc(ref) * scale + bias
    ref = np.maximum(ref, 0)
    y = eng.conv2d(torch.from_numpy(x).cuda(), w, stride=stride, rate=rate, pad_t=pad, pad_l=pad,
                   out_hw=ref.shape[1:3], scale=scale, bias=bias, relu=True).cpu().numpy()
    assert y.shape == ref.shape
    assert _rel_err(y, ref) < 2e-5


def test_conv_residual_and_subsample(eng):
    from oracle import dgp_oracle as O
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 10, 12, 64)).astype(np.float32)
    w = (rng.standard_normal((1, 1, 64, 256)) / 8).astype(np.float32)
    res_full = rng.standard_normal((2, 19, 24, 256)).astype(np.float32)      # subsampled by 2 -> 10 x 12
    ref = O._to_nhwc(O.conv2d(O._to_nchw(x), w, 1)) + res_full[:, ::2, ::2, :]
    ref = np.maximum(ref, 0)
    y = eng.conv2d(torch.from_numpy(x).cuda(), w, residual=torch.from_numpy(res_full).cuda(), res_stride=2,
                   relu=True).cpu().numpy()
    assert _rel_err(y, ref) < 2e-5
    res_same = rng.standard_normal((2, 10, 12, 256)).astype(np.float32)
    ref = O._to_nhwc(O.conv2d(O._to_nchw(x), w, 1)) + res_same
    y = eng.conv2d(torch.from_numpy(x).cuda(), w, residual=torch.from_numpy(res_same).cuda(), res_stride=1).cpu().numpy()
    assert _rel_err(y, ref) < 2e-5


@pytest.mark.parametrize("hw", [(240, 320), (187, 101), (6, 7)])
def test_maxpool_bit_exact(eng, hw):
    from oracle import dgp_oracle as O
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2,) + hw + (64,)).astype(np.float32)
    ref = O._to_nhwc(O.max_pool_same(O._to_nchw(x), 3, 2))
    y = eng.maxpool_3x3s2_same(torch.from_numpy(x).cuda()).cpu().numpy()
    assert y.shape == ref.shape
    assert np.array_equal(y, ref)


def test_preprocess_bit_exact(eng):
    from oracle import dgp_oracle as O
    rng = np.random.default_rng(2)
    f = rng.integers(0, 256, size=(2, 33, 35, 3), dtype=np.uint8)
    ref = f.astype(np.float32) - np.asarray(O.MEAN_PIXEL, dtype=np.float32)
    y = eng.preprocess_u8(torch.from_numpy(f).cuda()).cpu().numpy()
    assert np.array_equal(y[..., :3], ref)
    assert np.all(y[..., 3] == 0)


# ---------------------------------------------------------------------------- soft-argmax
def _peaky_scmap(rng, B, H, W, C, amp=8.0):
    s = rng.standard_normal((B, H, W, C)).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for b in range(B):
        for c in range(C):
            cy, cx = rng.uniform(0, H - 1), rng.uniform(0, W - 1)
            s[b, :, :, c] += amp * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / 8.0)
    return s


@pytest.mark.parametrize("shape,gl,th", [((3, 60, 80, 4), 1, 0.1), ((2, 30, 40, 20), 2, 0.5), ((1, 7, 5, 3), 1, 0.0),
                                         ((2, 9, 8, 2), 1, 1.0)])
def test_argmax_2d_from_cm_threshold_branch_matches_oracle(eng, shape, gl, th):
    """argmax_2d_from_cm(..., th=...) (fitdgp_util.py:377-388, no reference driver uses it): values below th * max of each map are
    zeroed, the map renormalised and the expectation retaken -- the drop-in's signature through the HIP kernels vs the oracle."""
    from oracle import dgp_oracle as O
    from deepgraphpose_amd.models.fitdgp_util import argmax_2d_from_cm
    rng = np.random.default_rng(11)
    s = _peaky_scmap(rng, *shape)
    mu_ref, pm_ref = O.argmax_2d_from_cm(s, 1.0, gl, th=th)
    mu64, pm64 = O.argmax_2d_from_cm(s, 1.0, gl, dtype=np.float64, th=th)
    mu, pmap = argmax_2d_from_cm(torch.from_numpy(s).cuda(), shape[3], 1.0, gl, th=th)
    mu, pmap = mu.cpu().numpy(), pmap.cpu().numpy()
    # an element within rounding of the cut may fall on either side in fp32: compare against fp64 where the fp32 oracle and fp64 agree
    # on the kept set, which holds for these seeds (asserted)
    assert np.array_equal(pm_ref > 0, pm64 > 0)
    assert np.array_equal(pmap > 0, pm_ref > 0)
    assert np.abs(pmap - pm_ref).max() < 1e-6
    assert np.abs(mu - mu_ref).max() * STRIDE < PX_TOL and np.abs(mu - mu64).max() * STRIDE < PX_TOL
    if th == 0.0:       # nothing is cut: the plain soft-argmax
        mu0, _ = argmax_2d_from_cm(torch.from_numpy(s).cuda(), shape[3], 1.0, gl)
        assert np.abs(mu - mu0.cpu().numpy()).max() < 1e-5
    if th == 1.0:       # only the maxima survive: the hard arg-max of the blurred map
        flat = pm_ref.transpose(0, 3, 1, 2).reshape(shape[0], shape[3], -1)
        am = np.stack(np.unravel_index(flat.argmax(-1), (shape[1], shape[2])), -1)
        assert np.abs(mu - am).max() < 1e-5


@pytest.mark.parametrize("shape,gl,gamma", [((3, 60, 80, 4), 1, 1.0), ((2, 94, 104, 5), 1, 1.0),
                                            ((1, 90, 160, 20), 2, 1.0), ((2, 60, 80, 4), 2, 0.5),
                                            ((2, 7, 5, 3), 1, 2.0), ((1, 1, 1, 1), 1, 1.0)])
def test_soft_argmax_matches_oracle(eng, shape, gl, gamma):
    from oracle import dgp_oracle as O
    rng = np.random.default_rng(7)
    s = _peaky_scmap(rng, *shape)
    mu_ref, pm_ref = O.argmax_2d_from_cm(s, gamma, gl)
    mu64, _ = O.argmax_2d_from_cm(s, gamma, gl, dtype=np.float64)
    mu, conf, idx, pmap = eng.soft_argmax(torch.from_numpy(s).cuda(), gamma, gl, want_pmap=True)
    mu, conf, idx, pmap = mu.cpu().numpy(), conf.cpu().numpy(), idx.cpu().numpy(), pmap.cpu().numpy()
    assert np.abs(mu - mu_ref).max() * STRIDE < PX_TOL
    assert np.abs(mu - mu64).max() * STRIDE < PX_TOL
    assert np.abs(pmap - pm_ref).max() < 1e-6
    for b in range(shape[0]):
        iref, lref = O.likelihood_window(s[b], mu[b])
        assert np.array_equal(idx[b], iref)
        assert np.abs(conf[b] - lref).max() < 2e-6


def test_soft_argmax_known_answers(eng):
    """One-hot scoremaps: closed-form expectation incl. the zero-padded border cells."""
    from oracle import dgp_oracle as O
    H, W = 12, 9
    g = O.gaussian_taps(1)
    np.testing.assert_allclose(g, [0.27406862, 0.45186276, 0.27406862], rtol=1e-6)
    for (r, c) in [(5, 4), (0, 0), (11, 8), (0, 4)]:
        s = np.full((1, H, W, 1), -200.0, dtype=np.float32)
        s[0, r, c, 0] = 0.0
        mu, conf, idx = eng.soft_argmax(torch.from_numpy(s).cuda(), 1.0, 1)
        def exp1(p, n):
            ks = [(p + d, g[d + 1]) for d in (-1, 0, 1) if 0 <= p + d < n]
            return sum(a * b for a, b in ks) / sum(b for _, b in ks)
        assert abs(mu[0, 0, 0].item() - exp1(r, H)) < 1e-5
        assert abs(mu[0, 0, 1].item() - exp1(c, W)) < 1e-5


def test_threshold_branch_on_large_maps_stays_within_the_gate(eng):
    """argmax_2d_from_cm's th branch on maps of ~50 000 cells (flat, and noisy with a broad softmax): a relative error d of the
    normaliser moves mu by d x mu, so the kernel sums in double -- with fp32 sums it was 1.3-2.1e-3 px from float64 here
    (found by scripts/fuzz_readout.py)."""
    from oracle import dgp_oracle as O
    rng = np.random.default_rng(5)
    for (H, W, C, gamma, gl, th, flat) in ((211, 254, 3, 10.45, 2, 0.30, True), (254, 222, 2, 2.63, 4, 0.13, True), (230, 240, 2, 1.0, 3, 0.45, False)):
        s = np.zeros((1, H, W, C), dtype=np.float32)
        if not flat:            # one strong peak per map far from the centre: the cells that survive the cut are the blur's taps around it
            s += (0.05 * rng.standard_normal(s.shape)).astype(np.float32)
            s[0, 201, 17, 0] += 14.0
            s[0, 9, 222, 1] += 14.0
        mu, conf, idx, pmap = eng.soft_argmax(torch.from_numpy(s).cuda(), gamma, gl, want_pmap=True)
        mu_t = eng.pmap_threshold(pmap.clone(), th).cpu().numpy()
        mu64, pm64 = O.argmax_2d_from_cm(s, gamma, gl, dtype=np.float64, th=th)
        _, pm = O.argmax_2d_from_cm(s, gamma, gl, dtype=np.float64)
        thr = pm.max(axis=(1, 2), keepdims=True) * th
        assert (np.abs(pm - thr) / thr).min() > 1e-5            # no cell sits on the threshold: the comparison is well defined
        assert np.abs(mu_t - mu64).max() * STRIDE < 2e-4, (H, W, float(np.abs(mu_t - mu64).max() * STRIDE))


def test_empty_batches_give_empty_outputs(eng):
    """No frames (the empty shard of a short video) or no joints: well-defined empty results, like sess.run on an empty feed --
    not an error about a null pointer (scripts/probe_edges.py walks the other degenerate arguments)."""
    from deepgraphpose_amd.synthetic import make_weights
    z = torch.zeros((0, 8, 8, 3), dtype=torch.float32, device="cuda")
    mu, conf, idx = eng.soft_argmax(z)
    assert mu.shape == (0, 3, 2) and conf.shape == (0, 3) and idx.shape == (0, 3, 2)
    assert eng.soft_argmax(torch.zeros((2, 8, 8, 0), device="cuda"))[0].shape == (2, 0, 2)
    assert eng.hard_argmax(z)[0].shape == (0, 3, 2) and eng.pmap_threshold(z, 0.5).shape == (0, 3, 2)
    net = eng.DGPNet(50, 3, 64, 96, max_batch=2)
    net.load_weights(make_weights(50, 3, False, seed=1, head_std=0.05))
    e = torch.zeros((0, 64, 96, 3), dtype=torch.uint8, device="cuda")
    assert net.infer(e)[0].shape == (0, 3, 2) and net.forward(e).shape == (0, 8, 12, 3)
    assert eng.motion_energy(torch.zeros((0, 8, 8, 3), dtype=torch.uint8, device="cuda")).shape == (0,)


def test_likelihood_tie_takes_first(eng):
    s = np.zeros((1, 6, 6, 1), dtype=np.float32)          # flat map: mu = centre 2.5, window 2x2, all tied
    mu, conf, idx = eng.soft_argmax(torch.from_numpy(s).cuda(), 1.0, 1)
    assert idx.cpu().numpy().tolist() == [[[2, 2]]]
    assert abs(conf.item() - 0.5) < 1e-7


@pytest.mark.parametrize("with_locref", [False, True])
def test_hard_argmax_bit_exact(eng, with_locref):
    from oracle import dgp_oracle as O
    rng = np.random.default_rng(11)
    B, H, W, C = 3, 60, 80, 4
    s = _peaky_scmap(rng, B, H, W, C)
    s[0, 10, 10, 0] = s[0, 40, 70, 0] = 50.0          # exact tie -> first in row-major order
    loc = rng.standard_normal((B, H, W, 2 * C)).astype(np.float32) if with_locref else None
    idx, prob, offs = eng.hard_argmax(torch.from_numpy(s).cuda(), None if loc is None else torch.from_numpy(loc).cuda())
    idx, prob, offs = idx.cpu().numpy(), prob.cpu().numpy(), offs.cpu().numpy()
    from deepgraphpose_amd.models.predict import pose_from_argmax
    for b in range(B):
        sig = O.sigmoid_f32(s[b])
        offmat = None if loc is None else loc[b].reshape(H, W, C, 2) * np.float32(7.2801)
        pose_ref, loc_ref = O.argmax_pose_predict(sig, offmat, 8.0)
        assert np.array_equal(idx[b], loc_ref)
        pose = pose_from_argmax(idx[b], prob[b], offs[b] if with_locref else None, 8.0, 7.2801)
        np.testing.assert_allclose(pose[:, :2], pose_ref[:, :2], atol=1e-4)
        np.testing.assert_allclose(pose[:, 2], pose_ref[:, 2], atol=2e-6)
    assert idx[0, 0].tolist() == [10, 10]


# ---------------------------------------------------------------------------- whole net
# (nj = 1 / 2: the heads' pointwise panel has 16 nj or 32 nj columns; below 64 it is padded to the narrowest cell tile -- round 4 found the
#  default engine returning garbage scoremaps for one or two bodyparts, which no test had covered)
@pytest.mark.parametrize("hw,depth,nj,B", [((96, 128), 50, 4, 3), ((75, 83), 50, 5, 2), ((64, 96), 101, 20, 1), ((96, 128), 50, 1, 2),
                                           ((64, 96), 50, 2, 3), ((64, 96), 50, 3, 1)])
def test_network_small_matches_oracle(eng, hw, depth, nj, B):
    from oracle import dgp_oracle as O
    from deepgraphpose_amd.synthetic import make_weights, make_frames
    wts = make_weights(depth, nj, True, seed=3, head_std=0.05)
    frames = make_frames(B, hw[0], hw[1], nj, seed=4)
    net = eng.DGPNet(depth, nj, hw[0], hw[1], max_batch=B, with_locref=True)
    net.load_weights(wts)
    scmap, locref, feats = net.forward(torch.from_numpy(frames).cuda(), want_locref=True, want_features=True)
    f_ref = O.resnet_features(frames, wts, depth)
    s_ref, l_ref = O.pose_heads(f_ref, wts, True)
    assert feats.shape == f_ref.shape and scmap.shape == s_ref.shape and locref.shape == l_ref.shape
    assert _rel_err(feats.cpu().numpy(), f_ref) < 1e-4
    assert _rel_err(scmap.cpu().numpy(), s_ref) < 1e-4
    assert _rel_err(locref.cpu().numpy(), l_ref) < 1e-4


@pytest.mark.parametrize("hw,depth", [((65, 97), 50), ((128, 160), 50), ((64, 96), 101)])
def test_slim_atrous_invariant_through_dgp_forward(eng, hw, depth):
    """TF-slim's own known answer for the atrous network (resnet_v1_test.py testAtrousFullyConvolutionalValues / testAtrousValuesBottleneck,
    restated on the oracle in tests/test_oracle_cpu.py): the output_stride = 16 network that DLC / DGP build (pose_net.py:46-52), subsampled
    by 2, IS slim's nominal stride-32 network.  dgp_forward only knows the stride-16 graph, so the invariant is checked ACROSS the two
    implementations: the engine's block4 features at the even pixels against the oracle's DENSE-FREE walk (output_stride = 32: no atrous
    conv, a strided 3x3 in block3's last unit) -- odd and even frame sizes, both backbones.  TF asserts atol = rtol = 1e-4 in fp32."""
    from oracle import dgp_oracle as O
    from deepgraphpose_amd.synthetic import make_weights, make_frames
    nj, B = 4, 2
    wts = make_weights(depth, nj, False, seed=21, head_std=0.05)
    frames = make_frames(B, hw[0], hw[1], nj, seed=22)
    net = eng.DGPNet(depth, nj, hw[0], hw[1], max_batch=B)
    net.load_weights(wts)
    _, feats = net.forward(torch.from_numpy(frames).cuda(), want_features=True)
    nominal = O.resnet_features(frames, wts, depth, output_stride=32)
    sub = feats.cpu().numpy()[:, ::2, ::2]
    assert sub.shape == nominal.shape
    assert _rel_err(sub, nominal) < 1e-4


def test_infer_640x480_r50_matches_oracle(eng):
    """BASELINE config 2 shapes (ResNet-50, 640x480, 4 joints), batch 2 so the oracle takes seconds."""
    from oracle import dgp_oracle as O
    from deepgraphpose_amd.synthetic import make_weights, make_frames
    nj, B = 4, 2
    wts = make_weights(50, nj, False, seed=0, head_std=0.05)
    frames = make_frames(B, 480, 640, nj, seed=0)
    net = eng.DGPNet(50, nj, 480, 640, max_batch=B)
    net.load_weights(wts)
    scm = torch.empty((B, 60, 80, nj), dtype=torch.float32, device="cuda")
    mu, conf, idx = net.infer(torch.from_numpy(frames).cuda(), gamma=1.0, gauss_len=1, scmap_out=scm)
    ref = O.infer(frames, wts, 50, STRIDE, 1.0, 1)
    mu = mu.cpu().numpy()
    x = mu[:, :, 1].astype(np.float64) * STRIDE + 0.5 * STRIDE
    y = mu[:, :, 0].astype(np.float64) * STRIDE + 0.5 * STRIDE
    err_px = np.sqrt((x - ref["x"]) ** 2 + (y - ref["y"]) ** 2)
    print("max px err", err_px.max(), "scmap rel err", _rel_err(scm.cpu().numpy(), ref["scmap"]))
    assert err_px.max() < PX_TOL
    assert np.array_equal(idx.cpu().numpy(), ref["idx"])
    assert np.abs(conf.cpu().numpy() - ref["likelihoods"]).max() < 1e-4


def test_demo_frame_size_747x832_five_joints_matches_oracle(eng):
    """BASELINE configs[0]'s shape on the HIP path: the Reaching demo's frames are 832 x 747 with 5 bodyparts
    (data/Reaching-Mackenzie-2018-08-30/config.yaml:6-11).  747 x 832 -> 374 x 416 -> 187 x 208 -> 94 x 104 -> 47 x 52: every stride-2
    stage halves an ODD size somewhere, so each SAME / conv2d_same padding asymmetry (pad 0 before / 1 after, explicit pad before the
    strided 3 x 3) is exercised at once.  Both heads, one frame, vs the oracle."""
    from oracle import dgp_oracle as O
    from deepgraphpose_amd.synthetic import make_weights, make_frames
    H, W, nj = 747, 832, 5
    wts = make_weights(50, nj, True, seed=11, head_std=0.05)
    frames = make_frames(1, H, W, nj, seed=12)
    net = eng.DGPNet(50, nj, H, W, max_batch=1, with_locref=True)
    net.load_weights(wts)
    assert (net.out_h, net.out_w, net.feat_h, net.feat_w) == (94, 104, 47, 52)
    ft = torch.from_numpy(frames).cuda()
    scmap, locref, feats = net.forward(ft, want_locref=True, want_features=True)
    f_ref = O.resnet_features(frames, wts, 50)
    s_ref, l_ref = O.pose_heads(f_ref, wts, True)
    assert feats.shape == f_ref.shape and scmap.shape == s_ref.shape == (1, 94, 104, nj) and locref.shape == l_ref.shape
    assert _rel_err(feats.cpu().numpy(), f_ref) < 1e-4
    assert _rel_err(scmap.cpu().numpy(), s_ref) < 1e-4
    assert _rel_err(locref.cpu().numpy(), l_ref) < 1e-4
    mu, conf, idx = net.infer(ft, gamma=1.0, gauss_len=1)
    ref = O.infer(frames, wts, 50, STRIDE, 1.0, 1)
    assert np.abs(mu.cpu().numpy() - ref["mu"]).max() * STRIDE < PX_TOL
    assert np.array_equal(idx.cpu().numpy(), ref["idx"])


def test_streaming_soft_argmax_is_bit_identical_to_the_lds_variant(eng):
    """Both instances of soft_argmax_kernel on the SAME maps (DGP_SOFTARGMAX_STREAM=1 forces the streaming one where the LDS one also fits):
    every output -- mu, likelihood, window index, pmap -- must be equal bit for bit, gamma != 1 included (the product s * gamma is rounded
    before the subtraction in both: __fmul_rn, no fma contraction), so a result cannot flip at the 38 400-cell boundary."""
    import os
    rng = np.random.default_rng(17)
    for (B, H, W, C, gl, gamma) in ((2, 60, 80, 4, 1, 1.0), (1, 192, 200, 2, 1, 0.37), (2, 33, 47, 5, 2, 2.5), (1, 1, 7, 1, 1, 1.3)):
        s = _peaky_scmap(rng, B, H, W, C) * np.float32(1.7)
        t = torch.from_numpy(s).cuda()
        lds = [x.clone() for x in eng.soft_argmax(t, gamma, gl, want_pmap=True)]
        os.environ["DGP_SOFTARGMAX_STREAM"] = "1"
        try:
            stream = eng.soft_argmax(t, gamma, gl, want_pmap=True)
        finally:
            del os.environ["DGP_SOFTARGMAX_STREAM"]
        for a, b, name in zip(lds, stream, ("mu", "conf", "idx", "pmap")):
            assert torch.equal(a, b), (name, (B, H, W, C, gl, gamma))


def test_maps_larger_than_the_lds_stream_from_global_memory(eng):
    """The reference's scoremap placeholders have no size limit ([None, None, None, nj], DGP/models/fitdgp.py:1130-1142).  dgp_soft_argmax keeps
    one joint's map in LDS up to 150 KB = 38 400 cells and STREAMS larger ones (softmax values recomputed from global memory where the blur
    reads them: the same expressions, a fallback not a fast path); dgp_infer therefore takes frames of any size.  (dgp_loss_fwd_bwd streams
    maps beyond its 19 200-cell LDS limit the same way since round 5: tests/test_train_gpu.py.)"""
    from deepgraphpose_amd import _lib
    from oracle import dgp_oracle as O
    rng = np.random.default_rng(5)
    for (H, W, C, gl) in ((200, 200, 2, 1), (192, 200, 2, 1), (150, 300, 3, 2)):       # 40 000 (streams), 38 400 (the largest LDS map), 45 000 (streams)
        s = (rng.standard_normal((2, H, W, C)) * 1.5).astype(np.float32)
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        for b in range(2):
            for j in range(C):
                py, px = rng.uniform(3, H - 4), rng.uniform(3, W - 4)
                s[b, :, :, j] += (14.0 * np.exp(-((yy - py) ** 2 + (xx - px) ** 2) / (2 * 2.0 ** 2))).astype(np.float32)
        mu, conf, idx, pmap = eng.soft_argmax(torch.from_numpy(s).cuda(), 1.0, gl, want_pmap=True)
        mu_ref, pm_ref = O.argmax_2d_from_cm(s, 1.0, gl)
        assert np.abs(mu.cpu().numpy() - mu_ref).max() * 8.0 < 1e-3
        assert np.abs(pmap.cpu().numpy() - pm_ref).max() <= 2e-6 * pm_ref.max() + 1e-12
        for b in range(2):
            iref, lref = O.likelihood_window(s[b], mu[b].cpu().numpy())
            assert np.array_equal(idx[b].cpu().numpy(), iref)
            np.testing.assert_allclose(conf[b].cpu().numpy(), lref, atol=2e-7)


def test_infer_on_frames_whose_scoremap_exceeds_the_lds(eng):
    """1616 x 1600 frames: scoremap 202 x 200 = 40 400 cells per joint, beyond the LDS variant of the soft-argmax; the whole path against the
    oracle (coordinates within 1e-3 px, window indices bit-exact)."""
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    from oracle import dgp_oracle as O
    wts = make_weights(50, 2, False, seed=9, head_std=0.05)
    frames = make_frames(1, 1616, 1600, 2, seed=10)
    net = eng.DGPNet(50, 2, 1616, 1600, max_batch=1)
    net.load_weights(wts)
    assert net.out_h * net.out_w > 38400
    mu, conf, idx = net.infer(torch.from_numpy(frames).cuda())
    ref = O.infer(frames, wts, 50, 8.0, 1.0, 1)
    assert np.abs(mu.cpu().numpy() - ref["mu"]).max() * 8.0 < 1e-3
    assert np.array_equal(idx.cpu().numpy(), ref["idx"])
    assert np.abs(conf.cpu().numpy() - ref["likelihoods"]).max() < 1e-5


# ---------------------------------------------------------------------------- golden vectors from the reference
def test_hip_kernels_reproduce_reference_vectors(eng):
    """A4 / A6 outputs of the reference's own numpy code (tests/golden/make_golden.py) from the HIP kernels."""
    import os
    from deepgraphpose_amd.models.predict import pose_from_argmax
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.npz"))
    for i in range(int(g["app_n"])):
        logits = g["app%d_logits" % i]
        H, W, Cn = logits.shape
        has_off = ("app%d_off" % i) in g.files
        loc = None
        if has_off:      # fixture offsets are already * locref_stdev; feed raw locref = off / stdev is lossy -> use stdev 1
            loc = torch.from_numpy(g["app%d_off" % i].reshape(1, H, W, 2 * Cn)).cuda()
        idx, prob, offs = eng.hard_argmax(torch.from_numpy(logits[None]).cuda(), loc)
        pose = pose_from_argmax(idx[0].cpu().numpy(), prob[0].cpu().numpy(), offs[0].cpu().numpy() if has_off else None,
                                8.0, 1.0)
        ref = g["app%d_pose" % i]
        np.testing.assert_array_equal(pose[:, :2], ref[:, :2])          # integer index + fp32 offset: exact
        np.testing.assert_allclose(pose[:, 2], ref[:, 2], atol=2e-7)
    from oracle import dgp_oracle as O
    for i in range(int(g["lik_n"])):
        # fixtures whose mu is a random point, not the scoremap's soft-argmax: the kernel computes its OWN mu, so the fixture's
        # (idx, lik) apply only where that mu falls into the fixture's window (same floor / ceil); otherwise the kernel is checked
        # against the oracle's restatement of the same lines (pinned to these fixtures in tests/test_oracle_cpu.py)
        s, mu_ref = g["lik%d_scmap" % i], g["lik%d_mu" % i]
        mu, conf, idx = eng.soft_argmax(torch.from_numpy(s[None]).cuda(), 1.0, 1)
        mu, conf, idx = mu[0].cpu().numpy(), conf[0].cpu().numpy(), idx[0].cpu().numpy()
        iref, lref = O.likelihood_window(s, mu)
        assert np.array_equal(idx, iref)
        np.testing.assert_allclose(conf, lref, atol=2e-7)
        same = np.all((np.floor(mu) == np.floor(mu_ref)) & (np.ceil(mu) == np.ceil(mu_ref)), axis=1)
        assert np.array_equal(idx[same], g["lik%d_idx" % i][same])
        np.testing.assert_allclose(conf[same], g["lik%d_lik" % i][same], atol=2e-7)
    # round 4: fixtures whose mu IS the soft-argmax of their scoremap -> the kernel's own (mu, idx, likelihood) against the
    # outputs of the reference's eval.py:329-343 lines DIRECTLY, every joint of every case (no oracle in between)
    n_direct = 0
    for i in range(int(g["likmu_n"])):
        s, mu_ref = g["likmu%d_scmap" % i], g["likmu%d_mu" % i]
        mu, conf, idx = eng.soft_argmax(torch.from_numpy(s[None]).cuda(), 1.0, int(g["likmu%d_gauss_len" % i]))
        np.testing.assert_allclose(mu[0].cpu().numpy(), mu_ref, atol=1e-3 / 8.0)             # 1e-3 px at stride 8
        np.testing.assert_array_equal(idx[0].cpu().numpy(), g["likmu%d_idx" % i])
        np.testing.assert_allclose(conf[0].cpu().numpy(), g["likmu%d_lik" % i], atol=2e-7)
        n_direct += mu_ref.shape[0]
    assert n_direct >= 32


def test_estimate_pose_end_to_end(eng, tmp_path):
    """estimate_pose (DGP/models/eval.py:217) over a frame stack: project layout in, DLC csv out, vs oracle."""
    import yaml
    from oracle import dgp_oracle as O
    from deepgraphpose_amd import weights_io
    from deepgraphpose_amd.models import eval as E
    from deepgraphpose_amd.synthetic import make_weights, make_frames
    nj, T = 3, 5
    parts = ["a", "b", "c"]
    proj = tmp_path / "proj"
    train = proj / "dlc-models" / "iteration-0" / "DemoOct2-trainset95shuffle1" / "train"
    train.mkdir(parents=True)
    (proj / "config.yaml").write_text(yaml.safe_dump(dict(Task="Demo", date="Oct2", iteration=0, TrainingFraction=[0.95],
                                                          bodyparts=parts, skeleton=[], project_path=str(proj))))
    (train / "pose_cfg.yaml").write_text(yaml.safe_dump(dict(num_joints=nj, all_joints_names=parts, net_type="resnet_50")))
    wts = make_weights(50, nj, False, seed=9, head_std=0.05)
    snap = weights_io.save_weights(str(train / "snapshot-step2-final--0"), wts)
    frames = make_frames(T, 96, 128, nj, seed=5)
    np.save(tmp_path / "clip.npy", frames)
    out = E.estimate_pose(str(proj / "config.yaml"), snap, str(tmp_path / "clip.npy"), str(tmp_path / "pred"),
                          shuffle=1, batch_size=2)
    ref = O.infer(frames, wts, 50, 8.0, 1.0, 1)
    assert np.abs(out["x"] - ref["x"]).max() < PX_TOL and np.abs(out["y"] - ref["y"]).max() < PX_TOL
    assert np.abs(out["likelihoods"] - ref["likelihoods"]).max() < 1e-4
    back = E.load_pose_from_dlc_to_dict(str(tmp_path / "pred" / "clip_labeled.csv"))
    np.testing.assert_allclose(back["x"], out["x"], rtol=1e-12)
    # second call: labels exist -> returns the csv path (eval.py:247-249)
    again = E.estimate_pose(str(proj / "config.yaml"), snap, str(tmp_path / "clip.npy"), str(tmp_path / "pred"))
    assert again.endswith("clip_labeled.csv")
    # sess.run drop-in
    cfg = E.yaml.safe_load(open(proj / "config.yaml"))
    from deepgraphpose_amd.config import get_train_config
    dlc_cfg = get_train_config(dict(cfg, video_path=None), shuffle=1)
    sess, mu_n, softmax, scmap, locref, inputs = E.setup_dgp_eval_graph(dlc_cfg, snap)
    mu_b, sc_b = sess.run([mu_n, scmap], feed_dict={inputs: frames[:1].astype(np.float32)})
    assert mu_b.shape == (1, nj, 2) and sc_b.shape == (1, 12, 16, nj)
    assert np.abs(mu_b[0] - ref["mu"][0]).max() * STRIDE < PX_TOL
    dlc_cfg.net_type = "resnet_101"
    with pytest.raises(KeyError):
        E.setup_dgp_eval_graph(dlc_cfg, snap)


def test_estimate_pose_on_the_16_bit_tier_beside_the_parity_tier(eng, tmp_path, monkeypatch):
    """The 16-bit tier behind the reference's entry point (round 6): estimate_pose(..., tier="f16") -- or DGP_EVAL_TIER=f16 in the
    environment -- runs the same host pipeline on H1 cells.  Same project, same frames, both tiers: the parity tier meets the north-star
    gate against the oracle (1e-3 px, likelihood window indices through the likelihoods), the 16-bit tier stays inside ITS reported band
    (0.08 px on ResNet-50, likelihoods within 0.05) -- and the two really are different engines."""
    import yaml
    from oracle import dgp_oracle as O
    from deepgraphpose_amd import weights_io
    from deepgraphpose_amd.models import eval as E
    from deepgraphpose_amd.synthetic import make_weights, make_frames
    nj, T = 4, 70
    parts = ["a", "b", "c", "d"]
    proj = tmp_path / "proj"
    train = proj / "dlc-models" / "iteration-0" / "DemoOct2-trainset95shuffle1" / "train"
    train.mkdir(parents=True)
    (proj / "config.yaml").write_text(yaml.safe_dump(dict(Task="Demo", date="Oct2", iteration=0, TrainingFraction=[0.95],
                                                          bodyparts=parts, skeleton=[], project_path=str(proj))))
    (train / "pose_cfg.yaml").write_text(yaml.safe_dump(dict(num_joints=nj, all_joints_names=parts, net_type="resnet_50")))
    wts = make_weights(50, nj, False, seed=9, head_std=0.05)
    snap = weights_io.save_weights(str(train / "snapshot-step2-final--0"), wts)
    frames = make_frames(T, 192, 256, nj, seed=5)
    ref = O.infer(frames, wts, 50, 8.0, 1.0, 1)
    monkeypatch.delenv("DGP_EVAL_TIER", raising=False)
    par = E.estimate_pose(str(proj / "config.yaml"), snap, frames, str(tmp_path / "p0"), save_pose=False, batch_size=16)
    f16 = E.estimate_pose(str(proj / "config.yaml"), snap, frames, str(tmp_path / "p1"), save_pose=False, batch_size=16, tier="f16")
    monkeypatch.setenv("DGP_EVAL_TIER", "f16")
    env = E.estimate_pose(str(proj / "config.yaml"), snap, frames, str(tmp_path / "p2"), save_pose=False, batch_size=16)
    monkeypatch.setenv("DGP_EVAL_TIER", "fp8")
    with pytest.raises(ValueError):
        E.estimate_pose(str(proj / "config.yaml"), snap, frames, str(tmp_path / "p3"), save_pose=False, batch_size=16)

    def px(a):
        return float(np.sqrt((a["x"] - ref["x"]) ** 2 + (a["y"] - ref["y"]) ** 2).max())
    print("estimate_pose vs oracle: parity %.3g px, f16 %.3g px (lik %.3g)" % (px(par), px(f16), np.abs(f16["likelihoods"] - ref["likelihoods"]).max()))
    assert px(par) < PX_TOL and np.abs(par["likelihoods"] - ref["likelihoods"]).max() < 1e-4
    assert px(f16) < 0.08 and np.abs(f16["likelihoods"] - ref["likelihoods"]).max() < 0.05
    assert px(f16) > 10 * px(par)                                    # (it IS the other arithmetic)
    for k in ("x", "y", "likelihoods"):
        assert np.array_equal(env[k], f16[k])                        # argument and environment select the same engine, deterministically


# fp32-MFMA tiles 0-6; bf16 6-term split 7 / 9 / 10 / 12; bf16 3-term split 8 / 11 (16-bit products: looser bound);
# fp16 high/low split 13-16 (need operand ranges)
_TILE_CASES = [(2, 19, 21, 64, 128, 3, 1, 1), (1, 15, 20, 128, 256, 3, 1, 2), (2, 20, 24, 256, 512, 1, 2, 1),
               (1, 30, 40, 1024, 256, 1, 1, 1), (2, 19, 21, 64, 64, 3, 2, 1)]
_TILES = [0, 1, 2, 4, 5, 6, 7, 9, 10, 12, 8, 11, 13, 14, 15, 16]
_TILES_128 = (0, 4, 5, 7, 8, 10, 11, 12, 13, 14, 16)                  # tiles that need Cout % 128 == 0


@pytest.mark.parametrize("case,tile", [(c, t) for c in _TILE_CASES for t in _TILES if not (t in _TILES_128 and c[4] % 128)])
def test_every_tile_variant_matches_oracle(eng, case, tile, monkeypatch):
    """All workgroup shapes of conv_igemm_f32 (incl. the 8-wave and loader-specialised variants) and of the split
    kernels conv_igemm_split_ls on the same layers, with residual + ReLU in the epilogue (combinations a tile cannot take --
    128-column tiles on the 64-channel layer -- are not generated)."""
    from oracle import dgp_oracle as O
    N, H, W, Cin, Cout, k, stride, rate = case
    monkeypatch.setenv("DGP_FORCE_TILE", str(tile))
    rng = np.random.default_rng(tile * 100 + Cin)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    scale = (1 + 0.1 * rng.standard_normal(Cout)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(Cout)).astype(np.float32)
    ref = O._to_nhwc(O.conv2d_same(O._to_nchw(x), w, stride, rate))
    res = rng.standard_normal(ref.shape).astype(np.float32)
    ref = np.maximum(ref * scale + bias + res, 0)
    keff = (k - 1) * rate + 1
    pad = (keff - 1) // 2 if stride > 1 else O.tf_same_pads(H, k, stride, rate)[1]
    y, yr = eng.conv2d(torch.from_numpy(x).cuda(), w, stride=stride, rate=rate, pad_t=pad, pad_l=pad, out_hw=ref.shape[1:3],
                       scale=scale, bias=bias, residual=torch.from_numpy(res).cuda(), res_stride=1, relu=True,
                       ranged=tile >= 13, return_range=True)
    y = y.cpu().numpy()
    assert _rel_err(y, ref) < (3e-4 if tile in (8, 11) else 2e-5)
    assert float(yr.max()) == float(np.abs(y).max())                # the epilogue's range tracking is exact


def test_fp16_split_keeps_fp32_class_accuracy_over_wide_ranges(eng, monkeypatch):
    """The fp16 high/low split scales both operands by powers of two taken from their measured maxima: results must stay
    fp32-class (vs float64) whatever the magnitudes -- tiny / huge weights and activations, one large outlier, zeros,
    negative values -- and never overflow."""
    rng = np.random.default_rng(7)
    N, H, W, Cin, Cout = 2, 12, 16, 128, 128
    monkeypatch.setenv("DGP_FORCE_TILE", "16")
    base = rng.standard_normal((N, H, W, Cin))
    wbase = rng.standard_normal((1, 1, Cin, Cout)) / np.sqrt(Cin)
    for xs, ws, outlier in ((1.0, 1.0, None), (1e-6, 1e-5, None), (3e4, 2e3, None), (1.0, 1.0, 1e4), (1e-3, 50.0, 7.0), (0.0, 1.0, None)):
        x = (base * xs).astype(np.float32)
        if outlier is not None:
            x[0, 0, 0, 0] = outlier
        w = (wbase * ws).astype(np.float32)
        ref = np.einsum("nhwc,co->nhwo", x.astype(np.float64), w[0, 0].astype(np.float64))
        y32 = eng.conv2d(torch.from_numpy(x).cuda(), w).cpu().numpy()                      # bf16x6 (no ranges)
        y16 = eng.conv2d(torch.from_numpy(x).cuda(), w, ranged=True).cpu().numpy()         # fp16 split
        assert np.isfinite(y16).all()
        # compare row-wise against what float32 inputs allow: |err| <= c * sum_k |x_k w_k| * 2^-22
        bound = np.einsum("nhwc,co->nhwo", np.abs(x).astype(np.float64), np.abs(w[0, 0]).astype(np.float64)) * 2.0 ** -21 + 1e-300
        if outlier is not None:     # elements far below the maximum keep >= 11 bits and an absolute error << max * 2^-30
            bound = bound + float(np.abs(x).max()) * float(np.abs(w).max()) * Cin * 2.0 ** -32
        assert (np.abs(y16 - ref) <= bound).all(), (xs, ws, outlier, float((np.abs(y16 - ref) / bound).max()))
        assert (np.abs(y32 - ref) <= bound).all()


@pytest.mark.parametrize("mode,extra", [("f32", {}), ("bf16x6", {}), ("f16x3", {"DGP_FUSE_SHORTCUT": "0", "DGP_STEM_ROWS": "0",
                                                                                "DGP_PRESPLIT_WEIGHTS": "0"}),
                                        ("f16x3", {"DGP_FUSE_SHORTCUT": "0"}),               # H2 engine, shortcut convs as their own launches
                                        ("f16x3", {"DGP_STEM_FUSED": "0"}),                  # H2 engine, root block as three launches
                                        ("f16x3", {"DGP_H2": "0"})])                         # fp32 activations (round-1 path)
def test_other_conv_modes_keep_parity(eng, mode, extra):
    """The conv path is chosen once per process (DGP_CONV_MODE and the A/B switches are read at first use), so the
    non-default paths -- fp32 MFMA, the range-free bf16x6 split, and fp16x3 without fused shortcut / row-walk stem /
    pre-split weights -- run in a child process: whole-network parity vs the oracle on a small ResNet-50."""
    import subprocess, sys, os, textwrap
    _skip_unless_tuning_build(extra)
    code = textwrap.dedent('''
        import sys, numpy as np, torch
        sys.path.insert(0, %r)
        from deepgraphpose_amd import engine, synthetic
        from oracle import dgp_oracle as O
        wts = synthetic.make_weights(50, 3, False, seed=2, head_std=0.05)
        fr = synthetic.make_frames(3, 96, 128, 3, seed=4)
        net = engine.DGPNet(50, 3, 96, 128, max_batch=3)
        net.load_weights(wts)
        mu, conf, idx = net.infer(torch.from_numpy(fr).cuda(), 1.0, 1)
        ref = O.infer(fr, wts, 50, 8.0, 1.0, 1)
        d = float(np.abs(mu.cpu().numpy() - ref["mu"]).max() * 8.0)
        ok = np.array_equal(idx.cpu().numpy(), ref["idx"])
        print("RESULT", d, ok)
    ''') % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DGP_CONV_MODE=mode, **extra)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    assert line, out.stdout[-2000:] + out.stderr[-2000:]
    _, d, ok = line[0].split()
    assert float(d) < PX_TOL and ok == "True", line[0]


def test_inference_net_changes_frame_size(eng):
    """dgp_net_set_input_size on an inference net: same weights (incl. the load-time derived panels), another geometry,
    parity at both sizes and back."""
    from oracle import dgp_oracle as O
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    nj = 3
    wts = make_weights(50, nj, False, seed=9, head_std=0.05)
    net = eng.DGPNet(50, nj, 96, 128, max_batch=2)
    net.load_weights(wts)
    for (h, w) in ((96, 128), (75, 109), (96, 128)):
        net.set_input_size(h, w)
        fr = make_frames(2, h, w, nj, seed=h)
        mu, conf, idx = net.infer(torch.from_numpy(fr).cuda(), 1.0, 1)
        ref = O.infer(fr, wts, 50, 8.0, 1.0, 1)
        assert mu.shape == (2, nj, 2) and (net.out_h, net.out_w) == tuple(ref["scmap"].shape[1:3]) if "scmap" in ref else True
        assert np.abs(mu.cpu().numpy() - ref["mu"]).max() * STRIDE < PX_TOL
        assert np.array_equal(idx.cpu().numpy(), ref["idx"])


def _pad_cin(x, w):
    """the conv kernel takes Cin = 4 * 2^k and Cout % 4 == 0: zero input channels / filter rows and zero filter columns up to that (the
    sums do not change; the callers read the first Cout columns back)"""
    c, co = x.shape[3], w.shape[3]
    cp = 4
    while cp < c:
        cp *= 2
    cop = (co + 3) // 4 * 4
    xp = np.zeros(x.shape[:3] + (cp,), np.float32); xp[..., :c] = x
    wp = np.zeros(w.shape[:2] + (cp, cop), np.float32); wp[:, :, :c, :co] = w
    return xp, wp


def test_tf_conv2d_known_answers_through_the_c_abi(eng):
    """TensorFlow's own conv2d known answers (conv_ops_test.py Conv2DTest, tests/_tf_kat.py) and TF-slim's conv2d_same ones
    (resnet_v1_test.py testConv2DSameEven / Odd) through the PRODUCT's conv kernel (dgp_conv2d over the C-ABI): HWIO filters, no kernel
    flip, VALID, TF's SAME (extra pixel after) and slim's explicit symmetric padding -- exact, the values are small integers."""
    import _tf_kat as K
    for name, in_shape, f_shape, stride, padding, expected in K.TF_CONV2D_KNOWN_ANSWERS:
        oh, pt, _ = K.tf_out_and_pads(in_shape[1], f_shape[0], stride, padding)
        ow, pl, _ = K.tf_out_and_pads(in_shape[2], f_shape[1], stride, padding)
        x, w = _pad_cin(K.tf_test_values(in_shape), K.tf_test_values(f_shape))
        y = eng.conv2d(torch.from_numpy(x).cuda(), w, stride, 1, pt, pl, out_hw=(oh, ow))
        assert y.shape[:3] == (in_shape[0], oh, ow), name
        np.testing.assert_array_equal(y.cpu().numpy()[..., :f_shape[3]].reshape(-1), np.asarray(expected, np.float32), err_msg=name)
        assert not y.cpu().numpy()[..., f_shape[3]:].any(), name
    for n, dense, strided_same, strided_tf in (
            (4, [[14, 28, 43, 26], [28, 48, 66, 37], [43, 66, 84, 46], [26, 37, 46, 22]], [[14, 43], [43, 84]], [[48, 37], [37, 22]]),
            (5, [[14, 28, 43, 58, 34], [28, 48, 66, 84, 46], [43, 66, 84, 102, 55], [58, 84, 102, 120, 64], [34, 46, 55, 64, 30]],
             [[14, 43, 34], [43, 84, 55], [34, 55, 30]], [[14, 43, 34], [43, 84, 55], [34, 55, 30]])):
        g = (np.arange(n).reshape(n, 1) + np.arange(n).reshape(1, n)).astype(np.float32)           # create_test_input: x[i, j] = i + j
        x, w = _pad_cin(g.reshape(1, n, n, 1), (np.arange(3).reshape(3, 1) + np.arange(3).reshape(1, 3)).astype(np.float32).reshape(3, 3, 1, 1))
        x = torch.from_numpy(x).cuda()
        y1 = eng.conv2d(x, w, 1, 1, 1, 1, out_hw=(n, n))                                           # slim.conv2d(stride 1, SAME)
        np.testing.assert_array_equal(y1.cpu().numpy()[0, :, :, 0], np.asarray(dense, np.float32))
        o = (n + 1) // 2
        y3 = eng.conv2d(x, w, 2, 1, 1, 1, out_hw=(o, o))                                           # resnet_utils.conv2d_same(stride 2): pad 1 before
        np.testing.assert_array_equal(y3.cpu().numpy()[0, :, :, 0], np.asarray(strided_same, np.float32))
        _, pt, _ = K.tf_out_and_pads(n, 3, 2, "SAME")
        y4 = eng.conv2d(x, w, 2, 1, pt, pt, out_hw=(o, o))                                         # slim.conv2d(stride 2, SAME): TF's own padding
        np.testing.assert_array_equal(y4.cpu().numpy()[0, :, :, 0], np.asarray(strided_tf, np.float32))


@pytest.mark.parametrize("tier", ["parity", "f16"])
@pytest.mark.parametrize("hw", [(75, 83), (64, 97), (61, 130)])
def test_root_block_reads_frames_at_any_byte_alignment(eng, hw, tier):
    """The root block fetches four pixels (12 bytes) per thread with one aligned 16-byte load: a frame batch that starts at any byte
    offset, rows of any length (W * 3 not a multiple of 4) and the batch's last bytes must give the bits of an aligned copy."""
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    h, w = hw
    wts = make_weights(50, 2, False, seed=12, head_std=0.05)
    fr = make_frames(3, h, w, 2, seed=13)
    net = eng.DGPNet(50, 2, h, w, max_batch=3, tier=tier)
    net.load_weights(wts)
    base = torch.from_numpy(fr).cuda()
    ref = [x.clone() for x in net.infer(base, 1.0, 1)]
    n = fr.size
    for delta in (1, 2, 3):
        buf = torch.zeros(n + 8, dtype=torch.uint8, device="cuda")
        buf[delta:delta + n] = base.reshape(-1)
        view = buf[delta:delta + n].view(3, h, w, 3)
        assert view.data_ptr() % 4 == (base.data_ptr() + delta) % 4
        got = net.infer(view, 1.0, 1)
        for a, b in zip(ref, got):
            assert torch.equal(a, b), (hw, tier, delta)


# ---------------------------------------------------------------------------- motion energy (8(f) N4)
@pytest.mark.parametrize("shape", [(1, 8, 8, 3), (5, 17, 23, 3), (40, 48, 64, 3), (19, 31, 37, 3), (33, 480, 640, 3)])
def test_motion_energy_is_bit_exact(eng, shape):
    """Integer work: the device sums must reproduce the reference's wrapped uint8 differences exactly, for frame sizes that
    are and are not multiples of the 16-byte load (byte-granular kernel), across the 16-frame groups and with a chained chunk."""
    from oracle import dgp_oracle as O
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    clip = rng.integers(0, 256, shape, dtype=np.uint8)
    clip[1:3] = clip[0]                                   # still frames -> exact zeros
    if shape[0] > 4:
        clip[4] = 255 - clip[3]                           # large wrapped differences
    want = O.motion_energy(clip)
    dev = torch.from_numpy(clip).cuda()
    got = eng.motion_energy(dev)
    assert got.dtype == np.float64 and np.array_equal(got, want)
    if shape[0] > 2:                                      # chunked: the second chunk continues from the first one's last frame
        k = shape[0] // 2
        a = eng.motion_energy(dev[:k])
        b = eng.motion_energy(dev[k:], prev=dev[k - 1])
        assert np.array_equal(np.concatenate([a, b]), want)


def test_calculate_motion_energy_hip_backend_matches_host(eng):
    from deepgraphpose_amd import dataset as D
    from deepgraphpose_amd.frames import ArraySource
    rng = np.random.default_rng(7)
    clip = rng.integers(0, 256, (70, 24, 32, 3), dtype=np.uint8)
    host = D.calculate_motion_energy(ArraySource(clip), backend="host")
    hip = D.calculate_motion_energy(ArraySource(clip), backend="hip", chunk=32)
    assert np.array_equal(host, hip)

    class FrameOnly:                                       # a source without iter_batches (moviepy-like)
        def iter_frames(self):
            yield from clip
    assert np.array_equal(D.calculate_motion_energy(FrameOnly(), backend="hip", chunk=16), host)


@pytest.mark.parametrize("extra", [{}, {"DGP_DMA": "0"}, {"DGP_COMPUTE_SPLIT": "0"}])
def test_cell_kernels_match_fp64_per_layer(eng, extra):
    """Single layers on the engine's pre-split-cell kernels (compute-side split, LDS-DMA loaders, 16x16x32 pipelined loop;
    DGP_CONV2D_CELLS=1 is read once per process, so this runs in a child): pointwise / 3x3 / dilated / strided shapes with ragged
    row counts and several K depths against an fp64 reference, plus an exact selection-matrix case that catches any row / column /
    k permutation (operands chosen so that every product is exact)."""
    import subprocess, sys, os, textwrap
    _skip_unless_tuning_build(extra)
    code = textwrap.dedent('''
        import sys, numpy as np, torch
        sys.path.insert(0, %r)
        from deepgraphpose_amd import engine
        rng = np.random.default_rng(11)
        worst = 0.0
        for (N, H, W, Cin, Cout, k, s, r) in [(1, 8, 16, 32, 128, 1, 1, 1), (1, 7, 11, 64, 128, 1, 1, 1), (1, 5, 9, 96 + 32, 256, 1, 1, 1),
                                              (2, 9, 13, 256, 128, 1, 1, 1), (1, 15, 20, 128, 256, 3, 1, 1),
                                              (1, 9, 11, 512, 512, 3, 1, 2), (2, 20, 24, 256, 512, 1, 2, 1), (3, 7, 9, 64, 128, 3, 2, 1),
                                              (1, 30, 40, 2048, 512, 1, 1, 1)]:
            x = np.maximum(rng.standard_normal((N, H, W, Cin)), 0).astype(np.float32) * rng.uniform(0.1, 3, (1, 1, 1, Cin)).astype(np.float32)
            w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
            keff = r * (k - 1) + 1
            Ho, Wo = -(-H // s), -(-W // s)
            ph, pw = max((Ho - 1) * s + keff - H, 0), max((Wo - 1) * s + keff - W, 0)          # TF SAME
            y = engine.conv2d(torch.from_numpy(x).cuda(), w, stride=s, rate=r, pad_t=ph // 2, pad_l=pw // 2, out_hw=(Ho, Wo),
                              ranged=True).cpu().numpy()
            xt = torch.from_numpy(x).double().permute(0, 3, 1, 2)
            wt = torch.from_numpy(w).double().permute(3, 2, 0, 1)
            xp = torch.nn.functional.pad(xt, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
            ref = torch.nn.functional.conv2d(xp, wt, stride=s, dilation=r).permute(0, 2, 3, 1).numpy()
            assert y.shape == ref.shape, (y.shape, ref.shape)
            worst = max(worst, float(np.abs(y - ref).max() / np.abs(ref).max()))
        # exact case: out[m][n] = x[m][n %% 32] with integer x < 2^13 (needs the high AND the low fp16 part)
        M, Cin, Cout = 128 + 37, 32, 256
        x = (np.arange(M)[:, None] * 64 + np.arange(Cin)[None, :]).astype(np.float32).reshape(1, 1, M, Cin)
        w = np.zeros((1, 1, Cin, Cout), np.float32)
        w[0, 0, np.arange(Cout) %% Cin, np.arange(Cout)] = 1.0
        y = engine.conv2d(torch.from_numpy(x).cuda(), w, ranged=True).cpu().numpy().reshape(M, Cout)
        exact = bool(np.array_equal(y, x.reshape(M, Cin)[:, np.arange(Cout) %% Cin]))
        print("RESULT", worst, exact)
    ''') % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DGP_CONV2D_CELLS="1", **extra)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    assert line, out.stdout[-2000:] + out.stderr[-2000:]
    _, worst, exact = line[0].split()
    assert float(worst) < 3e-6 and exact == "True", line[0]


@pytest.mark.parametrize("cells", ["0", "1"])
def test_full_size_layers_are_exactly_linear_in_powers_of_two(eng, cells):
    """Size-independent property at BASELINE's full layer sizes (batch 32, 30x40 maps), where the oracle is too slow: the fp16
    high/low split scales its operands by powers of two taken from the tracked ranges, so conv(4 x) must equal 4 conv(x) BIT FOR
    BIT (same mantissas in every partial product), and conv(x; 0.5 w) = 0.5 conv(x; w).  Child process: DGP_CONV2D_CELLS selects
    the register-staged split kernels (0) or the cell / LDS-DMA / 16x16x32 kernels (1)."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent('''
        import sys, numpy as np, torch
        sys.path.insert(0, %r)
        from deepgraphpose_amd import engine
        rng = np.random.default_rng(5)
        ok = True
        for (N, H, W, Cin, Cout, k, r, pad) in [(32, 30, 40, 512, 512, 3, 2, 2), (32, 30, 40, 2048, 512, 1, 1, 0), (32, 30, 40, 256, 1024, 1, 1, 0)]:
            x = torch.relu(torch.randn((N, H, W, Cin), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)))
            w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
            y1 = engine.conv2d(x, w, rate=r, pad_t=pad, pad_l=pad, out_hw=(H, W), ranged=True)
            y4 = engine.conv2d(x * 4.0, w, rate=r, pad_t=pad, pad_l=pad, out_hw=(H, W), ranged=True)
            yh = engine.conv2d(x, w * 0.5, rate=r, pad_t=pad, pad_l=pad, out_hw=(H, W), ranged=True)
            ok = ok and bool(torch.equal(y4, y1 * 4.0)) and bool(torch.equal(yh, y1 * 0.5)) and bool(torch.isfinite(y1).all())
        print("RESULT", ok)
    ''') % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DGP_CONV2D_CELLS=cells)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    assert line, out.stdout[-2000:] + out.stderr[-2000:]
    assert line[0].split()[1] == "True", line[0]


def test_full_size_inference_is_deterministic_and_batch_order_free(eng):
    """BASELINE configs[1] size (ResNet-50, 640x480, batch 32): two runs give bit-identical outputs (no float atomics anywhere: the
    grid-tail K-split and the heads sum their slices in fixed order), and reversing the frame order permutes the results -- exactly
    for the integer indices, within the parity tolerance for the coordinates (a frame's pixels land in other tiles, and a tail tile's
    K-split rounds differently from an unsplit one)."""
    from deepgraphpose_amd import synthetic
    B = 32
    net = eng.DGPNet(50, 4, 480, 640, max_batch=B)
    net.load_weights(synthetic.make_weights(50, 4, False, seed=0, head_std=0.05))
    f = torch.from_numpy(synthetic.make_frames(B, 480, 640, 4, seed=1)).cuda()
    mu1, c1, i1 = [t.clone() for t in net.infer(f, 1.0, 1)]
    mu2, c2, i2 = [t.clone() for t in net.infer(f, 1.0, 1)]
    assert torch.equal(mu1, mu2) and torch.equal(c1, c2) and torch.equal(i1, i2)
    mu3, c3, i3 = net.infer(torch.flip(f, dims=[0]).contiguous(), 1.0, 1)
    assert torch.equal(torch.flip(i3, dims=[0]), i1)
    assert float((torch.flip(mu3, dims=[0]) - mu1).abs().max()) * STRIDE < PX_TOL
    assert float((torch.flip(c3, dims=[0]) - c1).abs().max()) < 1e-4
