"""CPU tests of the oracle: golden vectors produced by the reference's own numpy code (A4, A6),
known answers for the TF-semantics restatement (A1-A3), and an independent C cross-check."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import dgp_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))


# ------------------------------------------------------------------ golden: reference numpy code
def test_argmax_pose_predict_matches_reference_vectors():
    for i in range(int(GOLD["app_n"])):
        scmap = GOLD["app%d_scmap" % i]
        off = GOLD["app%d_off" % i] if ("app%d_off" % i) in GOLD.files else None
        pose, loc = O.argmax_pose_predict(scmap, off, 8.0)
        np.testing.assert_array_equal(pose, GOLD["app%d_pose" % i])
        # the oracle's fp32 sigmoid reproduces the fixture's probabilities from the logits
        np.testing.assert_allclose(O.sigmoid_f32(GOLD["app%d_logits" % i]), scmap, rtol=0, atol=1.2e-7)
    assert GOLD["app0_pose"][0, :2].tolist() == [3 * 8 + 4 + float(GOLD["app0_off"][2, 3, 0, 0]),
                                                2 * 8 + 4 + float(GOLD["app0_off"][2, 3, 0, 1])]


def test_likelihood_window_matches_reference_vectors():
    for i in range(int(GOLD["lik_n"])):
        idx, lik = O.likelihood_window(GOLD["lik%d_scmap" % i], GOLD["lik%d_mu" % i])
        np.testing.assert_array_equal(idx, GOLD["lik%d_idx" % i])
        np.testing.assert_array_equal(lik, GOLD["lik%d_lik" % i])
    for i in range(int(GOLD["likmu_n"])):           # mu = the scoremap's own soft-argmax (round 4; the GPU test uses these directly)
        s = GOLD["likmu%d_scmap" % i]
        mu, _ = O.argmax_2d_from_cm(s[None], 1.0, int(GOLD["likmu%d_gauss_len" % i]))
        np.testing.assert_array_equal(mu[0], GOLD["likmu%d_mu" % i])
        idx, lik = O.likelihood_window(s, mu[0])
        np.testing.assert_array_equal(idx, GOLD["likmu%d_idx" % i])
        np.testing.assert_array_equal(lik, GOLD["likmu%d_lik" % i])


# ------------------------------------------------------------------ known answers (TF semantics)
def test_tf_same_padding_arithmetic():
    assert O.tf_same_pads(240, 3, 2) == (120, 0, 1)        # even: extra pixel AFTER
    assert O.tf_same_pads(187, 3, 2) == (94, 1, 1)
    assert O.tf_same_pads(30, 3, 1, 2) == (30, 2, 2)       # atrous rate 2
    assert O.tf_same_pads(60, 1, 2) == (30, 0, 0)


def test_feature_and_scoremap_dims():
    from deepgraphpose_amd.arch import feature_hw, scoremap_hw
    from deepgraphpose_amd.dataset import compute_pred_dims
    assert feature_hw(480, 640) == (30, 40) and scoremap_hw(480, 640) == (60, 80)
    assert feature_hw(747, 832) == (47, 52) and scoremap_hw(747, 832) == (94, 104)
    assert feature_hw(720, 1280) == (45, 80) and compute_pred_dims(720, 1280) == (90, 160)


def test_unit_plan_matches_slim_stack_blocks_dense():
    from deepgraphpose_amd.arch import resnet_units, conv_macs_per_frame
    u = resnet_units(50)
    assert len(u) == 16
    assert [x.stride for x in u] == [1, 1, 2, 1, 1, 1, 2] + [1] * 9        # stride on the LAST unit, block3 dense
    assert [x.rate for x in u] == [1] * 13 + [2, 2, 2]                      # block4 atrous rate 2
    assert len(resnet_units(101)) == 33
    assert abs(conv_macs_per_frame(480, 640) / 1e9 - 35.61) < 0.01         # BASELINE.md work model
    assert abs(conv_macs_per_frame(720, 1280, 101, 20, True) / 1e9 - 178.73) < 0.01


def test_oracle_unit_plan_is_independent_and_agrees_with_the_product():
    """The oracle walks slim's stack_blocks_dense itself (oracle/resnet_plan.py) -- it must not import the product's table --
    and the two independently written plans agree unit by unit for both backbones."""
    import inspect
    from oracle import resnet_plan, dgp_oracle, dgp_train_oracle
    from deepgraphpose_amd.arch import resnet_units
    for mod in (resnet_plan, dgp_oracle, dgp_train_oracle):
        assert "deepgraphpose_amd.arch import resnet_units" not in inspect.getsource(mod)
    for depth, n in ((50, 16), (101, 33)):
        a, b = resnet_plan.units(depth), resnet_units(depth)
        assert len(a) == len(b) == n
        for x, y in zip(a, b):
            assert (x.scope, x.depth_in, x.depth, x.depth_bottleneck, x.stride, x.rate, x.has_shortcut_conv) == \
                   (y.scope, y.depth_in, y.depth, y.depth_bottleneck, y.stride, y.rate, y.has_shortcut_conv)
    u = resnet_plan.units(50)
    assert [x.stride for x in u] == [1, 1, 2, 1, 1, 1, 2] + [1] * 9 and [x.rate for x in u] == [1] * 13 + [2, 2, 2]
    assert [x.has_shortcut_conv for x in u] == [True, False, False, True, False, False, False, True] + [False] * 5 + [True, False, False]


def test_gaussian_taps_constants():
    np.testing.assert_allclose(O.gaussian_taps(1), [0.27406862, 0.45186276, 0.27406862], rtol=1e-6)
    g2 = O.gaussian_taps(2)
    assert len(g2) == 5 and abs(g2.sum() - 1) < 1e-6 and g2[2] == g2.max()


def test_deconv_unit_impulse_stamps_kernel():
    """x = delta at (i, j) -> the 3x3 kernel appears at rows 2i..2i+2, cols 2j..2j+2 (no flip), cropped."""
    w = np.arange(9, dtype=np.float32).reshape(3, 3, 1, 1) + 1
    x = np.zeros((1, 4, 5, 1), np.float32)
    x[0, 1, 2, 0] = 1
    y = O.conv2d_transpose_same(x, w, None)[0, :, :, 0]
    assert y.shape == (8, 10)
    ref = np.zeros((8, 10), np.float32)
    ref[2:5, 4:7] = w[:, :, 0, 0]
    np.testing.assert_array_equal(y, ref)
    x[:] = 0
    x[0, 3, 4, 0] = 1                           # last cell: the stamp is cut by the 2H x 2W crop
    y = O.conv2d_transpose_same(x, w, None)[0, :, :, 0]
    np.testing.assert_array_equal(y[6:8, 8:10], w[:2, :2, 0, 0])


def test_soft_argmax_one_hot_closed_form():
    g = O.gaussian_taps(1).astype(np.float64)
    H, W = 10, 7
    for r, c in [(4, 3), (0, 0), (9, 6)]:
        s = np.full((1, H, W, 1), -200.0, np.float32)
        s[0, r, c, 0] = 0
        mu, pm = O.argmax_2d_from_cm(s, 1.0, 1)
        def ex(p, n):
            ks = [(p + d, g[d + 1]) for d in (-1, 0, 1) if 0 <= p + d < n]
            return sum(a * b for a, b in ks) / sum(b for _, b in ks)
        assert abs(mu[0, 0, 0] - ex(r, H)) < 1e-5 and abs(mu[0, 0, 1] - ex(c, W)) < 1e-5
        assert abs(pm.sum() - 1) < 1e-5


def test_argmax_tie_is_first_row_major():
    s = np.zeros((4, 5, 1), np.float32)
    s[1, 3, 0] = s[2, 0, 0] = 0.9
    _, loc = O.argmax_pose_predict(s, None, 8.0)
    assert loc.tolist() == [[1, 3]]


# ------------------------------------------------------------------ independent C restatement
@pytest.fixture(scope="module")
def naive():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libdgp_naive.so"))
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("k,stride,rate", [(1, 1, 1), (3, 1, 1), (3, 2, 1), (3, 1, 2), (7, 2, 1), (1, 2, 1)])
def test_conv_restatement_vs_naive_c(naive, k, stride, rate):
    rng = np.random.default_rng(k * 10 + stride + rate)
    N, H, W, Ci, Co = 2, 11, 14, 5, 6
    x = rng.standard_normal((N, H, W, Ci)).astype(np.float32)
    w = rng.standard_normal((k, k, Ci, Co)).astype(np.float32)
    ref = O._to_nhwc(O.conv2d_same(O._to_nchw(x), w, stride, rate))
    keff = (k - 1) * rate + 1
    pad = (keff - 1) // 2 if stride > 1 else O.tf_same_pads(H, k, 1, rate)[1]
    Ho, Wo = ref.shape[1:3]
    y = np.empty((N, Ho, Wo, Co), np.float32)
    naive.naive_conv2d(_p(x), N, H, W, Ci, _p(w), k, k, Co, stride, rate, pad, pad, Ho, Wo, _p(y))
    np.testing.assert_allclose(ref, y, rtol=1e-4, atol=1e-4)


def test_deconv_maxpool_softargmax_vs_naive_c(naive):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 5, 6, 7)).astype(np.float32)
    w = rng.standard_normal((3, 3, 4, 7)).astype(np.float32)
    b = rng.standard_normal(4).astype(np.float32)
    ref = O.conv2d_transpose_same(x, w, b)
    y = np.empty_like(ref)
    naive.naive_conv2d_transpose(_p(x), 2, 5, 6, 7, _p(w), 3, 3, 4, _p(b), 2, _p(y))
    np.testing.assert_allclose(ref, y, rtol=1e-4, atol=1e-4)
    for hw in [(12, 10), (11, 9)]:
        xp = rng.standard_normal((1,) + hw + (3,)).astype(np.float32)
        refp = O._to_nhwc(O.max_pool_same(O._to_nchw(xp), 3, 2))
        yp = np.empty_like(refp)
        naive.naive_maxpool_same(_p(xp), 1, hw[0], hw[1], 3, 3, 2, _p(yp))
        np.testing.assert_array_equal(refp, yp)
    s = (rng.standard_normal((2, 9, 8, 3)) * 3).astype(np.float32)
    for gl in (1, 2):
        mu32, _ = O.argmax_2d_from_cm(s, 1.5, gl)
        mu = np.empty((2, 3, 2), np.float64)
        naive.naive_soft_argmax(_p(s), 2, 9, 8, 3, C.c_double(1.5), gl, _p(mu))
        np.testing.assert_allclose(mu32, mu, atol=2e-5)


def test_batch_norm_and_backbone_shapes():
    from deepgraphpose_amd.synthetic import make_weights, make_frames
    w = make_weights(50, 3, True, seed=5)
    fr = make_frames(1, 64, 80, 3, seed=1)
    f, ends = O.resnet_features(fr, w, 50, return_endpoints=True)
    assert f.shape == (1, 4, 5, 2048)
    assert ends["conv1"].shape == (1, 32, 40, 64) and ends["pool1"].shape == (1, 16, 20, 64)
    s, l = O.pose_heads(f, w, True)
    assert s.shape == (1, 8, 10, 3) and l.shape == (1, 8, 10, 6)
    assert np.isfinite(f).all() and f.min() >= 0


def test_argmax_2d_threshold_branch_independent_restatement():
    """th branch of argmax_2d_from_cm (fitdgp_util.py:377-388) against a per-map loop written from the reference's text: st < max*th
    -> 0, renormalise, expectation over the (row, col) grid."""
    rng = np.random.default_rng(2)
    s = (rng.standard_normal((2, 9, 8, 3)) * 3).astype(np.float32)
    for th in (0.05, 0.5):
        mu, pm = O.argmax_2d_from_cm(s, 1.0, 1, dtype=np.float64, th=th)
        _, p0 = O.argmax_2d_from_cm(s, 1.0, 1, dtype=np.float64)
        for b in range(2):
            for c in range(3):
                st = p0[b, :, :, c].copy()
                st[st < st.max() * th] = 0.0
                st /= st.sum() + 1e-100
                np.testing.assert_allclose(pm[b, :, :, c], st, rtol=1e-12, atol=0)
                rows, cols = np.mgrid[0:9, 0:8]
                np.testing.assert_allclose(mu[b, c], [(st * rows).sum(), (st * cols).sum()], rtol=1e-12)
        assert (pm == 0).sum() > 0


def test_temporal_flow_weight_autograd_matches_finite_differences():
    """The temporal clique's flow weight (fitdgp.py:1085-1118) as a differentiable function of the marker positions: values equal the
    numpy restatement; the gradient (TF: CropAndResizeGradBoxes through min / max / the frame clamps / min(., 1)^3) matches central
    differences and vanishes where the mean flow in the box is <= 1."""
    import torch
    from oracle import dgp_train_oracle as T
    rng = np.random.default_rng(0)
    nt, nj, Hin, Win = 4, 3, 48, 64
    yy, xx = np.mgrid[0:Hin, 0:Win]
    vf = np.stack([np.abs(np.sin(xx / 9.0 + t) * np.cos(yy / 7.0)) * sc for t, sc in zip(range(nt - 1), (0.3, 2.5, 4.0))])
    P = torch.tensor(rng.uniform(5, 40, (nt, nj, 2)), dtype=torch.float64, requires_grad=True)
    wb = np.array([50.0, 50.0, 50.0])
    w = T.temporal_flow_weights_torch(P, vf, wb, 6, 8)
    np.testing.assert_allclose(w.detach().numpy(), T.temporal_flow_weights(P.detach().numpy(), vf, wb, 6, 8), rtol=0, atol=1e-14)
    w.sum().backward()
    g = P.grad.numpy()
    assert np.all(g[0] == 0)                          # frame 0 only touches pair 0, whose mean flow is < 1: weight clamped at 1
    assert np.abs(g[1:]).max() > 1e-3
    eps = 1e-5
    for (t, j, k) in ((2, 1, 0), (1, 2, 1), (3, 0, 0)):
        Pp = P.detach().clone().numpy(); Pm = Pp.copy()
        Pp[t, j, k] += eps; Pm[t, j, k] -= eps
        fd = (T.temporal_flow_weights(Pp, vf, wb, 6, 8).sum() - T.temporal_flow_weights(Pm, vf, wb, 6, 8).sum()) / (2 * eps)
        assert abs(fd - g[t, j, k]) < 1e-6 * max(1.0, abs(fd)), (t, j, k, fd, g[t, j, k])


# ------------------------------------------------------------------ TF / TF-slim's OWN published known answers
# The conv / pool / deconv arithmetic of the reference lives in TensorFlow 1.x + tf.contrib.slim (not vendored, not importable here).
# What IS public are the known answers TensorFlow's own unit tests assert; the tests below restate their inputs and expected outputs
# (source file and test name in each docstring) and check the oracle against them.  They pin the padding / alignment / stride
# bookkeeping of the restatement to numbers TF itself is held to -- the part of "TF semantics" that is not plain multiply-add.
def _slim_create_test_input(batch, height, width, channels):
    """tensorflow/contrib/slim/python/slim/nets/resnet_v1_test.py `create_test_input`: x[b, i, j, c] = i + j"""
    g = (np.arange(height).reshape(height, 1) + np.arange(width).reshape(1, width)).astype(np.float32)
    return np.tile(g.reshape(1, height, width, 1), [batch, 1, 1, channels])


def _nchw(a):
    return torch.from_numpy(np.ascontiguousarray(a)).permute(0, 3, 1, 2).contiguous()


def _hw(t):
    return t[0, 0].numpy()


def test_slim_subsample_known_answers():
    """resnet_v1_test.py ResnetUtilsTest.testSubsampleThreeByThree / testSubsampleFourByFour: range(9) as 3x3 -> [0, 2, 6, 8];
    range(16) as 4x4 -> [0, 2, 8, 10] (resnet_utils.subsample = 1x1 max-pool of stride 2: keeps the EVEN rows / columns)."""
    x = torch.arange(9, dtype=torch.float32).reshape(1, 1, 3, 3)
    assert O.subsample(x, 2).reshape(-1).tolist() == [0, 2, 6, 8]
    x = torch.arange(16, dtype=torch.float32).reshape(1, 1, 4, 4)
    assert O.subsample(x, 2).reshape(-1).tolist() == [0, 2, 8, 10]
    assert O.subsample(x, 1) is x


def test_slim_conv2d_same_even_known_answers():
    """resnet_v1_test.py ResnetUtilsTest.testConv2DSameEven: x = create_test_input(1, 4, 4, 1), 3x3 kernel w[a, b] = a + b, bias 0.
    y1 = slim.conv2d(stride 1, SAME); y2 = subsample(y1, 2); y3 = resnet_utils.conv2d_same(stride 2) == y2 (that is the POINT of
    conv2d_same: explicit symmetric padding, so the strided conv sits on the even pixels of the dense one); y4 = slim.conv2d(stride 2,
    SAME) -- TF's own SAME padding puts the extra pixel AFTER, so it sits on the ODD pixels and differs."""
    x = _nchw(_slim_create_test_input(1, 4, 4, 1))
    w = _slim_create_test_input(1, 3, 3, 1).reshape(3, 3, 1, 1)
    y1 = O.conv2d(x, w, 1, 1, "SAME")
    np.testing.assert_array_equal(_hw(y1), [[14, 28, 43, 26], [28, 48, 66, 37], [43, 66, 84, 46], [26, 37, 46, 22]])
    np.testing.assert_array_equal(_hw(O.subsample(y1, 2)), [[14, 43], [43, 84]])
    np.testing.assert_array_equal(_hw(O.conv2d_same(x, w, 2)), [[14, 43], [43, 84]])
    np.testing.assert_array_equal(_hw(O.conv2d(x, w, 2, 1, "SAME")), [[48, 37], [37, 22]])


def test_slim_conv2d_same_odd_known_answers():
    """resnet_v1_test.py ResnetUtilsTest.testConv2DSameOdd: the same on a 5x5 input -- here TF's SAME and conv2d_same agree (y4 == y2)."""
    x = _nchw(_slim_create_test_input(1, 5, 5, 1))
    w = _slim_create_test_input(1, 3, 3, 1).reshape(3, 3, 1, 1)
    y1 = O.conv2d(x, w, 1, 1, "SAME")
    np.testing.assert_array_equal(_hw(y1), [[14, 28, 43, 58, 34], [28, 48, 66, 84, 46], [43, 66, 84, 102, 55],
                                            [58, 84, 102, 120, 64], [34, 46, 55, 64, 30]])
    y2 = [[14, 43, 34], [43, 84, 55], [34, 55, 30]]
    np.testing.assert_array_equal(_hw(O.subsample(y1, 2)), y2)
    np.testing.assert_array_equal(_hw(O.conv2d_same(x, w, 2)), y2)
    np.testing.assert_array_equal(_hw(O.conv2d(x, w, 2, 1, "SAME")), y2)


def test_tf_conv2d_transpose_same_known_answer():
    """tensorflow/python/kernel_tests/conv2d_transpose_test.py Conv2DTransposeTest.testConv2DTransposeSame: x = ones [1, 6, 4, 3],
    filter = ones [3, 3, 2, 3] (HW, out, in), strides 2, output [1, 12, 8, 2], padding SAME.  Expected: 3.0 everywhere, + 3.0 where ONE
    of (h, w) is even and interior (0 < h < 11, 0 < w < 7), + 9.0 where both are -- i.e. input pixel i stamps the kernel at output rows
    2 i .. 2 i + 2, cropped to 2 H: the alignment the part_pred / locref_pred heads rely on (pose_net.py:18-26)."""
    x = np.ones((1, 6, 4, 3), np.float32)
    f = np.ones((3, 3, 2, 3), np.float32)
    y = O.conv2d_transpose_same(x, f, None, stride=2)
    assert y.shape == (1, 12, 8, 2)
    want = np.full((12, 8), 3.0, np.float32)
    for h in range(12):
        for w_ in range(8):
            h_in = h % 2 == 0 and 0 < h < 11
            w_in = w_ % 2 == 0 and 0 < w_ < 7
            if h_in and w_in:
                want[h, w_] += 9.0
            elif h_in or w_in:
                want[h, w_] += 3.0
    for k in range(2):
        np.testing.assert_array_equal(y[0, :, :, k], want)


def test_slim_stack_blocks_dense_stride_bookkeeping():
    """resnet_v1_test.py ResnetCompleteNetworkTest.testFullyConvolutionalEndpointShapes / testAtrousFullyConvolutionalEndpointShapes /
    testClassificationShapes: on a 321x321 input the block endpoints of a resnet_v1 with block strides (2, 2, 2, 1) are 41 / 21 / 11 / 11
    pixels wide at the nominal stride and 41 / 41 / 41 / 41 at output_stride = 8; on 224x224: 28 / 14 / 7 / 7.  ResNet-50 has those block
    strides, so its plan (oracle/resnet_plan.py, the stack_blocks_dense walk) must give those widths, and 41 / 21 / 21 / 21 at the
    output_stride = 16 that DLC builds (pose_net.py:49)."""
    from oracle import resnet_plan

    def widths(n, output_stride):
        n = -(-n // 2)                                                    # conv1: conv2d_same 7x7 / 2 -> ceil(n / 2)
        n = O.tf_same_pads(n, 3, 2)[0]                                    # pool1: 3x3 / 2 SAME
        out = {}
        for u in resnet_plan.units(50, output_stride):
            n = -(-n // u.stride)                                         # conv2 (conv2d_same) and the shortcut (subsample / 1x1 conv) agree
            out[u.scope.split("/")[1]] = n
        return [out["block%d" % b] for b in (1, 2, 3, 4)]

    assert widths(321, 32) == [41, 21, 11, 11]
    assert widths(321, 8) == [41, 41, 41, 41]
    assert widths(224, 32) == [28, 14, 7, 7]
    assert widths(321, 16) == [41, 21, 21, 21]
    rates = {os_: [u.rate for u in resnet_plan.units(50, os_)] for os_ in (8, 16, 32)}
    assert rates[32] == [1] * 16
    assert rates[16] == [1] * 13 + [2] * 3
    assert rates[8] == [1] * 7 + [2] * 6 + [4] * 3
    with pytest.raises(ValueError):
        resnet_plan.units(50, 6)                                          # "The output_stride needs to be a multiple of 4."


def _small_weights(depth, nj, seed):
    from deepgraphpose_amd.synthetic import make_weights
    return make_weights(depth, nj, False, seed=seed, head_std=0.05)


def test_slim_atrous_invariant_on_the_oracle():
    """resnet_v1_test.py ResnetCompleteNetworkTest.testAtrousFullyConvolutionalValues (and ResnetUtilsTest.testAtrousValuesBottleneck):
    "dense feature extraction followed by subsampling gives identical results to feature extraction at the nominal network output
    stride" -- output_stride = 16 features subsampled by 2 == output_stride = 32 features, odd AND even input sizes; TF asserts
    atol = rtol = 1e-4 on fp32.  In float64 the oracle's two walks agree to rounding; in fp32 within TF's own tolerance (relative to
    the feature range, which is O(1) here as in TF's test)."""
    wts = _small_weights(50, 4, 11)
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, size=(1, 65, 97, 3), dtype=np.uint8)          # odd height, odd width
    for shape in ((1, 65, 97, 3), (1, 64, 96, 3)):
        fr = frames[:, :shape[1], :shape[2]]
        d16 = O.resnet_features(fr, wts, 50, dtype=np.float64, output_stride=16)
        d32 = O.resnet_features(fr, wts, 50, dtype=np.float64, output_stride=32)
        assert d16.shape[1:3] == (-(-shape[1] // 16), -(-shape[2] // 16)) and d32.shape[1:3] == (-(-shape[1] // 32), -(-shape[2] // 32))
        sub = d16[:, ::2, ::2]
        assert sub.shape == d32.shape
        scale = np.abs(d32).max()
        assert np.abs(sub - d32).max() <= 1e-12 * scale
        f16_ = O.resnet_features(fr, wts, 50, output_stride=16)
        f32_ = O.resnet_features(fr, wts, 50, output_stride=32)
        assert np.abs(f16_[:, ::2, ::2] - f32_).max() <= 1e-4 * scale + 1e-4


@pytest.mark.parametrize("case", __import__("_tf_kat").TF_CONV2D_KNOWN_ANSWERS, ids=lambda c: c[0])
def test_tf_conv2d_known_answers(case):
    """tensorflow/python/kernel_tests/conv_ops_test.py Conv2DTest (tests/_tf_kat.py holds the vectors): the published expected outputs agree
    with the definition evaluated in float64 loops (so a mis-remembered vector cannot pin anything), and the oracle's conv2d -- HWIO
    filters, no kernel flip, VALID / SAME with the extra pixel AFTER -- reproduces them exactly."""
    import _tf_kat as K
    name, in_shape, f_shape, stride, padding, expected = case
    np.testing.assert_array_equal(K.brute_force(in_shape, f_shape, stride, padding).reshape(-1), np.asarray(expected, np.float64))
    y = O.conv2d(_nchw(K.tf_test_values(in_shape)), K.tf_test_values(f_shape), stride, 1, padding)
    np.testing.assert_array_equal(y.permute(0, 2, 3, 1).reshape(-1).numpy(), np.asarray(expected, np.float32))


def test_tf_max_pool_same_known_answer():
    """tensorflow/python/kernel_tests/pooling_ops_test.py PoolingTest._testMaxPoolSamePadding: input 1 .. 18 as [1, 2, 3, 3] (NHWC), 2 x 2 window,
    stride 2, SAME -> [13, 14, 15, 16, 17, 18]: the odd width is padded AFTER (the second window holds column 2 alone) and padding never wins.
    The root block's pool (slim.max_pool2d([3, 3], stride 2, 'SAME')) is the same operator with another window."""
    x = np.arange(1, 19, dtype=np.float32).reshape(1, 2, 3, 3)
    y = O.max_pool_same(_nchw(x), 2, 2)
    assert tuple(y.shape) == (1, 3, 1, 2)
    assert y.permute(0, 2, 3, 1).reshape(-1).tolist() == [13.0, 14.0, 15.0, 16.0, 17.0, 18.0]
    # 3 x 3 / 2 SAME by hand on an even and an odd size (out = ceil(n / 2)): n = 4 pads 0 before / 1 after -> windows {0, 1, 2}, {2, 3};
    # n = 5 pads 1 / 1 -> windows {0, 1}, {1, 2, 3}, {3, 4}
    for n, want in ((4, [[10, 11], [14, 15]]), (5, [[6, 8, 9], [16, 18, 19], [21, 23, 24]])):
        z = np.arange(n * n, dtype=np.float32).reshape(1, n, n, 1)
        np.testing.assert_array_equal(O.max_pool_same(_nchw(z), 3, 2)[0, 0].numpy(), want)
