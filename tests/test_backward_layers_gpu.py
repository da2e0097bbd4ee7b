"""Layer-level parity of the training step's weight-gradient and data-gradient kernels at the REAL shapes of config 4
(ResNet-50, 640x480, 11 frames), the full-size step against the autograd oracle, and the ResNet-101 / 20-joint /
dense-skeleton shapes of config 5.

The forward kernels have per-layer float64 tests in test_parity_gpu.py; these are the same for the backward kernels
(`wgrad_h3`, `wgrad_f32<1|2>`, the data-gradient convs incl. the zero-stuffed stride-2 gather, ReLU gate and shortcut adds).
Reference arithmetic: float64 im2col + matmul written out below (torch is only the float64 calculator).
Tolerance: 1e-5 relative (max-abs error over max |reference|) -- fp32 accumulation noise of a 13 200-pixel reduction.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5


def _same_pads(n, k, stride, rate):
    """conv2d_same (stride > 1: explicit symmetric-ish padding, VALID) / TF SAME (stride 1): (pad_before, out)."""
    keff = (k - 1) * rate + 1
    if stride > 1:
        tot = keff - 1
        pb = tot // 2
        return pb, (n + tot - keff) // stride + 1
    tot = keff - 1
    return tot // 2, n


def _im2col64(x, k, stride, rate, pad_t, pad_l, Ho, Wo):
    """x [N,H,W,C] float64 (device) -> [N*Ho*Wo, k*k*C] with zero padding; tap-major, channel-minor (HWIO rows)."""
    N, H, W, C = x.shape
    keff = (k - 1) * rate + 1
    Hp = (Ho - 1) * stride + keff
    Wp = (Wo - 1) * stride + keff
    xp = torch.zeros((N, max(Hp, H + pad_t), max(Wp, W + pad_l), C), dtype=x.dtype, device=x.device)
    xp[:, pad_t:pad_t + H, pad_l:pad_l + W] = x
    cols = []
    for kh in range(k):
        for kw in range(k):
            cols.append(xp[:, kh * rate: kh * rate + (Ho - 1) * stride + 1: stride,
                           kw * rate: kw * rate + (Wo - 1) * stride + 1: stride])
    return torch.stack(cols, 3).reshape(N * Ho * Wo, k * k * C)


# (N, H, W, Cin, Cout, k, stride, rate) -- the config-4 layer shapes (nt = 11)
WGRAD_CASES = [
    (11, 30, 40, 512, 512, 3, 1, 2),        # block4 conv2 (dilated): wgrad_h3, 128 x 128 tiles, 1024-workgroup grid
    (11, 30, 40, 1024, 256, 1, 1, 1),       # block3 conv1
    (11, 30, 40, 2048, 512, 1, 1, 1),       # block4 conv1 (deepest K)
    (11, 60, 80, 128, 128, 3, 2, 1),        # block2 unit_4 conv2: stride 2, conv2d_same padding
    (11, 60, 80, 256, 512, 1, 2, 1),        # block2 shortcut-like strided 1x1
    (11, 120, 160, 256, 64, 1, 1, 1),       # block1 conv1: Cout = 64 -> wgrad_f32<1>
    (11, 120, 160, 64, 64, 3, 1, 1),        # block1 conv2: Cout = 64
    (3, 480, 640, 4, 64, 7, 2, 1),          # stem (4-channel padded input)
]


@pytest.mark.parametrize("ranged", [True, False])
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad_layer_matches_float64(lib_built, case, ranged):
    from deepgraphpose_amd import engine
    N, H, W, Cin, Cout, k, stride, rate = case
    g = torch.Generator(device="cuda").manual_seed(hash(case) % (2 ** 31))
    pad_t, Ho = _same_pads(H, k, stride, rate)
    pad_l, Wo = _same_pads(W, k, stride, rate)
    x = torch.relu(torch.randn((N, H, W, Cin), generator=g, device="cuda"))          # post-ReLU activations
    dy = torch.randn((N, Ho, Wo, Cout), generator=g, device="cuda") * 1e-3
    dy[torch.rand((N, Ho, Wo, Cout), generator=g, device="cuda") < 0.4] = 0.0          # gated gradient
    dw, cs = engine.conv2d_wgrad(x, dy, k, stride, rate, pad_t, pad_l, ranged=ranged)
    cols = _im2col64(x.double(), k, stride, rate, pad_t, pad_l, Ho, Wo)
    ref = (cols.t() @ dy.double().reshape(-1, Cout)).reshape(k, k, Cin, Cout)
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    assert err < REL_TOL, (case, ranged, err)
    cref = dy.double().reshape(-1, Cout).sum(0)
    assert float((cs.double() - cref).abs().max() / (cref.abs().max() + 1e-30)) < REL_TOL


# the LDS-DMA tile of the training step (wgrad_dma): 128 x 128 layers only
DMA_CASES = [c for c in WGRAD_CASES if c[4] >= 128 and c[3] * c[5] * c[5] >= 128] + [
    (11, 30, 40, 256, 1024, 1, 1, 1),       # block3 conv3
    (11, 60, 80, 128, 512, 1, 1, 1),        # block2 conv3 (one k-tile)
    (3, 13, 17, 64, 136, 3, 1, 1),          # ragged everything: Wo < 16 (several row wraps per step), Cout % 128 != 0, M % 16 != 0, two taps per k-tile
    (2, 9, 11, 32, 128, 3, 2, 1),           # stride 2, four taps per k-tile, K = 288 (the last k-tile is part empty)
]


_DMA_RATIOS = [(1.0, 1.0), (0.4, 3.0), (2.0 ** -6, 1.0), (1.0, 2.0 ** 8), (0.0, 1.0)]


# (failed-prediction ratios: one large and one ragged shape only)
@pytest.mark.parametrize("case,ratio", [(c, r) for c in DMA_CASES for r in _DMA_RATIOS if r == (1.0, 1.0) or c in (DMA_CASES[0], DMA_CASES[-2])])
def test_wgrad_dma_tile_matches_float64(lib_built, case, ratio):
    """Both operands as fp16 high / low copies with predicted scales.  ratio = (previous / current maximum) of (x, dy): inside the
    usable window the LDS-DMA path runs, outside it (or with no previous range) the fp32-MFMA path of the same kernel -- same tolerance."""
    from deepgraphpose_amd import engine
    N, H, W, Cin, Cout, k, stride, rate = case
    g = torch.Generator(device="cuda").manual_seed(hash(case) % (2 ** 31))
    pad_t, Ho = _same_pads(H, k, stride, rate)
    pad_l, Wo = _same_pads(W, k, stride, rate)
    x = torch.relu(torch.randn((N, H, W, Cin), generator=g, device="cuda"))
    dy = torch.randn((N, Ho, Wo, Cout), generator=g, device="cuda") * 1e-3
    dy[torch.rand((N, Ho, Wo, Cout), generator=g, device="cuda") < 0.4] = 0.0
    dw, cs = engine.conv2d_wgrad_shadow(x, dy, k, stride, rate, pad_t, pad_l, prev_ratio=ratio)
    cols = _im2col64(x.double(), k, stride, rate, pad_t, pad_l, Ho, Wo)
    ref = (cols.t() @ dy.double().reshape(-1, Cout)).reshape(k, k, Cin, Cout)
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    assert err < REL_TOL, (case, ratio, err)
    cref = dy.double().reshape(-1, Cout).sum(0)
    assert float((cs.double() - cref).abs().max() / (cref.abs().max() + 1e-30)) < REL_TOL


def test_wgrad_dma_tile_small_values_keep_their_bits(lib_built):
    """A tensor whose bulk sits 2^-12 below its maximum (one outlier sets the scale): the copies' low pieces go subnormal there, the
    products still carry >= 18 bits and the sum over 13 200 pixels stays inside the layer tolerance."""
    from deepgraphpose_amd import engine
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.relu(torch.randn((11, 30, 40, 256), generator=g, device="cuda")) * 2.0 ** -12
    x[0, 0, 0, 0] = 3.0
    dy = torch.randn((11, 30, 40, 256), generator=g, device="cuda") * 1e-7
    dy[3, 4, 5, 6] = 2e-4
    dw, _ = engine.conv2d_wgrad_shadow(x, dy, 1)
    ref = (x.double().reshape(-1, 256).t() @ dy.double().reshape(-1, 256)).reshape(1, 1, 256, 256)
    assert float((dw.double() - ref).abs().max() / ref.abs().max()) < REL_TOL


def _dgrad_ref(dy, w, H, W, stride, rate, pad_t, pad_l):
    """float64: dx[n, h, w, ci] = sum over taps / co of dy[n, ho, wo, co] * w[kh, kw, ci, co] with h = ho*s + kh*r - pad."""
    N, Ho, Wo, Cout = dy.shape
    k, _, Cin, _ = w.shape
    keff = (k - 1) * rate + 1
    Hp = max((Ho - 1) * stride + keff, H + pad_t)
    Wp = max((Wo - 1) * stride + keff, W + pad_l)
    dxp = torch.zeros((N, Hp, Wp, Cin), dtype=torch.float64, device=dy.device)
    dyf = dy.reshape(-1, Cout)
    for kh in range(k):
        for kw in range(k):
            t = (dyf @ w[kh, kw].t()).reshape(N, Ho, Wo, Cin)
            dxp[:, kh * rate: kh * rate + (Ho - 1) * stride + 1: stride,
                kw * rate: kw * rate + (Wo - 1) * stride + 1: stride] += t
    return dxp[:, pad_t:pad_t + H, pad_l:pad_l + W]


DGRAD_CASES = [
    # N, H, W, Cin, Cout, k, stride, rate, add_mode (0 none, 1 same grid, -2 coarser grid)
    (11, 30, 40, 512, 512, 3, 1, 2, 0),
    (11, 30, 40, 1024, 256, 1, 1, 1, 1),       # conv1 of an identity unit: + shortcut gradient, gated by the unit input
    (11, 30, 40, 512, 2048, 1, 1, 1, 0),       # conv3
    (11, 60, 80, 128, 128, 3, 2, 1, 0),        # stride-2 conv2: dy read on the zero-stuffed grid
    (11, 60, 80, 256, 512, 1, 2, 1, 0),        # strided shortcut conv
    (11, 60, 80, 512, 128, 1, 1, 1, -2),       # conv1 of the strided unit: + subsample-shortcut gradient from the coarse grid
    (11, 120, 160, 256, 64, 1, 1, 1, 1),
    (11, 120, 160, 64, 64, 3, 1, 1, 0),
]


@pytest.mark.parametrize("ranged", [True, False, "h2gate"])
@pytest.mark.parametrize("case", DGRAD_CASES)
def test_dgrad_layer_matches_float64(lib_built, case, ranged):
    """ranged "h2gate": the ReLU gate tensor arrives as H2 cells (fast pass of the training step)."""
    from deepgraphpose_amd import engine
    N, H, W, Cin, Cout, k, stride, rate, add_mode = case
    mask_h2 = ranged == "h2gate"
    ranged = bool(ranged)
    g = torch.Generator(device="cuda").manual_seed(hash(case) % (2 ** 31))
    pad_t, Ho = _same_pads(H, k, stride, rate)
    pad_l, Wo = _same_pads(W, k, stride, rate)
    dy = torch.randn((N, Ho, Wo, Cout), generator=g, device="cuda") * 1e-3
    dy[torch.rand((N, Ho, Wo, Cout), generator=g, device="cuda") < 0.4] = 0.0
    w = torch.randn((k, k, Cin, Cout), generator=g, device="cuda") / float(np.sqrt(k * k * Cin))
    scale = 1.0 + 0.1 * torch.randn(Cout, generator=g, device="cuda")
    mask = torch.relu(torch.randn((N, H, W, Cin), generator=g, device="cuda"))
    add = None
    if add_mode == 1:
        add = torch.randn((N, H, W, Cin), generator=g, device="cuda") * 1e-3
    elif add_mode == -2:
        add = torch.randn((N, (H + 1) // 2, (W + 1) // 2, Cin), generator=g, device="cuda") * 1e-3
    dx = engine.conv2d_dgrad(dy, w, (H, W), stride, rate, pad_t, pad_l, scale=scale, mask=mask, dx_add=add,
                             add_mode=add_mode if add is not None else 1, ranged=ranged, mask_h2=mask_h2)
    ref = _dgrad_ref(dy.double(), (w * scale).double(), H, W, stride, rate, pad_t, pad_l)
    if add_mode == 1:
        ref = ref + add.double()
    elif add_mode == -2:
        ref[:, ::2, ::2] += add.double()
    ref = torch.where(mask > 0, ref, torch.zeros_like(ref))
    err = float((dx.double() - ref).abs().max() / ref.abs().max())
    assert err < REL_TOL, (case, ranged, err)


# ---------------------------------------------------------------------------------------------------------------
def _loss_cfg(hy, nj, S0, ws, ws_max, n_tot, n_vis):
    return dict(nj=nj, S0=S0, ws=ws, ws_max=ws_max, stride=8.0, gamma=hy.gamma, gauss_len=hy.gauss_len,
                lengthscale=hy.lengthscale, gm2=hy.gm2, gm3=hy.gm3, wn_visible=hy.wn_visible, wn_hidden=hy.wn_hidden,
                locref_loss_weight=hy.locref_loss_weight, locref_huber_loss=True, n_frames_total=n_tot,
                n_visible_frames_total=n_vis)


def _dense_skeleton(nj):
    pairs = [(a, b) for a in range(nj) for b in range(a + 1, nj)]
    S0 = np.zeros((len(pairs), nj))
    for l, (a, b) in enumerate(pairs):
        S0[l, a], S0[l, b] = 1, -1
    return S0


def test_loss_dense_skeleton_20_joints_matches_autograd(lib_built):
    """BASELINE configs[4] loss shape: 20 keypoints, dense skeleton (all 190 pairs), 90 x 160 scoremaps, gm2=1 gm3=3."""
    from test_train_gpu import _make_loss_case
    from deepgraphpose_amd.loss import dgp_loss_fwd_bwd, DGPHyper
    from oracle import dgp_train_oracle as T
    nt, H, W, nj = 4, 90, 160, 20
    rng = np.random.default_rng(20)
    batch, _ = _make_loss_case(rng, nt, H, W, nj, 2, 0.15, 2)
    S0 = _dense_skeleton(nj)
    assert S0.shape == (190, nj)
    pred = (rng.standard_normal((nt, H, W, nj)) * 2).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for n in range(nt):
        for j in range(nj):
            cy, cx = rng.uniform(0, H - 1), rng.uniform(0, W - 1)
            pred[n, :, :, j] += 6 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / 6.0)
    loc = rng.standard_normal((nt, H, W, 2 * nj)).astype(np.float32)
    hy = DGPHyper(gm2=1, gm3=3)
    ws, ws_max = rng.uniform(5, 20, 190), rng.uniform(10, 400, 190)
    n_tot, n_vis = 800.0, 41.0
    pt = torch.tensor(pred, dtype=torch.float64, requires_grad=True)
    lt = torch.tensor(loc, dtype=torch.float64, requires_grad=True)
    L = T.dgp_loss(pt, lt, batch, _loss_cfg(hy, nj, S0, ws, ws_max, n_tot, n_vis))
    L["total_loss"].backward()
    losses, dpred, dloc, mu = dgp_loss_fwd_bwd(torch.from_numpy(pred).cuda(), torch.from_numpy(loc).cuda(), batch, hy, S0, ws,
                                               ws_max, n_tot, n_vis)
    for k in ("visible_loss_pred", "hidden_loss_pred", "visible_loss_locref", "ws_loss", "total_loss"):
        assert abs(losses[k] - float(L[k])) <= 2e-5 * max(1.0, abs(float(L[k]))), (k, losses[k], float(L[k]))
    np.testing.assert_allclose(mu.cpu().numpy(), L["_mu"].detach().numpy(), atol=2e-5)
    gp = pt.grad.numpy()
    assert np.abs(dpred.cpu().numpy() - gp).max() <= 2e-4 * np.abs(gp).max() + 1e-9
    gl = lt.grad.numpy()
    assert np.abs(dloc.cpu().numpy() - gl).max() <= 2e-5 * np.abs(gl).max() + 1e-10


def test_resnet101_trainer_dense_skeleton_matches_autograd(lib_built):
    """Trainer(101, nj = 20) with the dense 190-limb skeleton: loss and every gradient vs torch autograd (float64 oracle)."""
    from test_train_gpu import _make_loss_case, _oracle_grads
    from deepgraphpose_amd.synthetic import make_weights, make_frames
    from deepgraphpose_amd.arch import scoremap_hw
    from deepgraphpose_amd.train import Trainer
    from deepgraphpose_amd.loss import DGPHyper
    nj, nt, hw = 20, 3, (64, 96)
    rng = np.random.default_rng(101)
    H, W = scoremap_hw(*hw)
    batch, _ = _make_loss_case(rng, nt, H, W, nj, 1, 0.1, 2)
    S0 = _dense_skeleton(nj)
    wts = make_weights(101, nj, True, seed=11, head_std=0.05)
    frames = make_frames(nt, hw[0], hw[1], nj, seed=11)
    ws, ws_max = rng.uniform(5, 20, 190), rng.uniform(10, 40, 190)
    hy = DGPHyper(gm2=1, gm3=3)
    n_tot, n_vis = 300.0, 25.0
    P, L = _oracle_grads(wts, frames, batch, S0, ws, ws_max, hy, n_tot, n_vis, depth=101, dtype=torch.float64)
    P32, _ = _oracle_grads(wts, frames, batch, S0, ws, ws_max, hy, n_tot, n_vis, depth=101, dtype=torch.float32)
    tr = Trainer(101, nj, hw[0], hw[1], max_frames=nt)
    tr.load_weights(wts)
    losses = tr.forward_backward(torch.from_numpy(frames).cuda(), batch, hy, S0, ws, ws_max, n_tot, n_vis)
    ref_total = float(L["total_loss"].detach())
    assert abs(losses["total_loss"] - ref_total) < 1e-4 * max(1, abs(ref_total))
    g = tr.get_grads()
    rel, rel32, tot_ref, tot_err = {}, {}, 0.0, 0.0
    for k, t in P.items():
        if not t.requires_grad:
            continue
        ref = t.grad.numpy()
        d = g[k].reshape(ref.shape) - ref
        nref = np.linalg.norm(ref.ravel()) + 1e-30
        rel[k] = np.linalg.norm(d.ravel()) / nref
        rel32[k] = np.linalg.norm((P32[k].grad.numpy().astype(np.float64) - ref).ravel()) / nref      # fp32 arithmetic's own error
        tot_ref += float((ref ** 2).sum())
        tot_err += float((d ** 2).sum())
    # With 60 hidden maps some scoremap peaks saturate (c = max sigmoid -> 1) and the gm3 weight (1 - c) loses digits in ANY fp32
    # evaluation (1e-3 relative here, the same in the fp32 CPU oracle): the bound for the tensors downstream of the loss gradient
    # only (block4 + heads: no upstream ReLU-gate flips) is therefore round-off (2e-5) plus three times the fp32 oracle's own
    # distance from the float64 truth, tensor by tensor.
    strict = {k: v for k, v in rel.items() if "block4" in k or k.startswith("pose/")}
    worst = sorted(((v / (2e-5 + 3 * rel32[k]), k, v, rel32[k]) for k, v in strict.items()), reverse=True)[:4]
    assert worst[0][0] < 1.0, worst
    assert rel["pose/locref_pred/block4/weights"] < 2e-5 and rel["pose/locref_pred/block4/biases"] < 2e-5     # no (1 - c) in this branch
    assert max(rel.values()) < 1e-2, sorted(rel.items(), key=lambda kv: -kv[1])[:4]       # single ReLU gate flips upstream, see test_train_gpu
    assert np.sqrt(tot_err / tot_ref) < 3e-3


def test_full_size_config4_step_matches_autograd(lib_built):
    """BASELINE configs[3] at FULL size: ResNet-50, 640 x 480, 11 frames (1 labeled + 10 unlabeled), gm2=1 gm3=3, chain skeleton.
    Loss and gradients of one step vs the fp32 autograd oracle on the host cores (the layer tests above pin each kernel to 1e-5;
    this pins their composition at the real grid sizes: pixel-slice splits, tail K-split, 1024-workgroup grids)."""
    from test_train_gpu import _make_loss_case, _oracle_grads
    from deepgraphpose_amd.synthetic import make_weights, make_frames
    from deepgraphpose_amd.train import Trainer
    from deepgraphpose_amd.loss import DGPHyper
    torch.set_num_threads(min(32, torch.get_num_threads() * 2 if torch.get_num_threads() < 16 else 32))
    nj, nt, hw = 4, 11, (480, 640)
    rng = np.random.default_rng(44)
    batch, _ = _make_loss_case(rng, nt, 60, 80, nj, 1, 0.0, 2)
    S0 = np.zeros((3, nj))
    for l in range(3):
        S0[l, l], S0[l, l + 1] = 1, -1
    wts = make_weights(50, nj, True, seed=4, head_std=0.05)
    frames = make_frames(nt, hw[0], hw[1], nj, seed=4)
    ws, ws_max = rng.uniform(5, 20, 3), rng.uniform(10, 40, 3)
    hy = DGPHyper(gm2=1, gm3=3)
    n_tot, n_vis = 1000.0, 50.0
    tr = Trainer(50, nj, hw[0], hw[1], max_frames=nt)
    tr.load_weights(wts)
    ft = torch.from_numpy(frames).cuda()
    losses = tr.forward_backward(ft, batch, hy, S0, ws, ws_max, n_tot, n_vis)
    g = tr.get_grads()
    assert all(np.isfinite(v) for v in losses.values())
    assert all(np.isfinite(v).all() for v in g.values())
    P, L = _oracle_grads(wts, frames, batch, S0, ws, ws_max, hy, n_tot, n_vis, depth=50, dtype=torch.float32)
    ref_total = float(L["total_loss"].detach())
    assert abs(losses["total_loss"] - ref_total) < 2e-4 * max(1, abs(ref_total)), (losses["total_loss"], ref_total)
    tot_ref = tot_err = 0.0
    rel = {}
    for k, t in P.items():
        if not t.requires_grad:
            continue
        ref = t.grad.numpy().astype(np.float64)
        d = g[k].reshape(ref.shape).astype(np.float64) - ref
        rel[k] = np.linalg.norm(d.ravel()) / (np.linalg.norm(ref.ravel()) + 1e-30)
        tot_ref += float((ref ** 2).sum())
        tot_err += float((d ** 2).sum())
    # fp32 on both sides: block4 + heads see no upstream ReLU-gate flips and agree to accumulation noise; everything else within
    # the gate-flip budget of test_full_backward_matches_autograd; global gradient direction to 1e-3
    strict = {k: v for k, v in rel.items() if "block4" in k or k.startswith("pose/")}
    assert max(strict.values()) < 5e-4, sorted(strict.items(), key=lambda kv: -kv[1])[:4]
    assert np.sqrt(tot_err / tot_ref) < 1e-3, np.sqrt(tot_err / tot_ref)
    # second pass on the same inputs: float atomics reorder, nothing else changes
    losses2 = tr.forward_backward(ft, batch, hy, S0, ws, ws_max, n_tot, n_vis)
    assert abs(losses2["total_loss"] - losses["total_loss"]) <= 1e-6 * max(1.0, abs(losses["total_loss"]))
    g2 = tr.get_grads()
    n1 = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in g.values()))
    n2 = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in g2.values()))
    assert abs(n1 - n2) <= 1e-5 * n1


# ---------------------------------------------------------------------------------------------------------------
def test_resnet101_1280x720_20_joints_matches_oracle(lib_built):
    """BASELINE configs[4] per-GPU shape: ResNet-101, 1280 x 720, 20 keypoints: one frame vs the CPU oracle (coordinates within
    1e-3 px, window indices bit-exact), then a 16-frame batch twice (bit-identical) whose first frame matches the oracle too."""
    from deepgraphpose_amd.engine import DGPNet
    from deepgraphpose_amd.synthetic import make_weights, make_frames
    from oracle import dgp_oracle as O
    nj = 20
    wts = make_weights(101, nj, False, seed=21)
    frames = make_frames(16, 720, 1280, nj, seed=21)
    ref = O.infer(frames[:1], wts, depth=101)
    net = DGPNet(101, nj, 720, 1280, max_batch=16, with_locref=False)
    net.load_weights(wts)
    ft = torch.from_numpy(frames).cuda()
    mu1, conf1, idx1 = [t.cpu().numpy() for t in net.infer(ft[:1])]
    assert np.abs(mu1 - ref["mu"]).max() * 8.0 < 1e-3
    assert np.array_equal(idx1, ref["idx"])
    a = [t.clone() for t in net.infer(ft)]
    b = [t.clone() for t in net.infer(ft)]
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert np.abs(a[0][:1].cpu().numpy() - ref["mu"]).max() * 8.0 < 1e-3
    assert np.array_equal(a[2][:1].cpu().numpy(), ref["idx"])


def test_likelihood_window_saturated_logit_matches_numpy(lib_built):
    """Logits above 88.7 make the reference's e^x / (e^x + 1) NaN in fp32 and np.argmax returns the FIRST NaN of the window
    (eval.py:337-343).  The kernel follows: index of the first NaN, likelihood NaN."""
    from deepgraphpose_amd import engine
    from oracle import dgp_oracle as O
    H, W, C = 12, 16, 3
    s = np.full((1, H, W, C), -4.0, dtype=np.float32)
    s[0, 5, 7, 0] = 95.0                    # the peak itself saturates: window = that cell (+ neighbours)
    s[0, 5, 8, 0] = 94.0
    s[0, 6, 7, 1] = 30.0                    # large but finite
    s[0, 3, 3, 2] = 91.0
    s[0, 4, 4, 2] = 91.0                    # two saturated cells: mu lands between them
    mu, conf, idx = engine.soft_argmax(torch.from_numpy(s).cuda(), 1.0, 1, want_pmap=False)
    mu, conf, idx = mu.cpu().numpy(), conf.cpu().numpy(), idx.cpu().numpy()
    with np.errstate(over="ignore", invalid="ignore"):
        iref, lref = O.likelihood_window(s[0], mu[0])
    assert np.array_equal(idx[0], iref)
    assert np.array_equal(np.isnan(conf[0]), np.isnan(lref))
    assert np.isnan(conf[0, 0])
    ok = ~np.isnan(lref)
    assert np.abs(conf[0][ok] - lref[ok]).max() < 2e-6


def test_full_size_backward_is_stable_across_repeats(lib_built):
    """Race check for the weight gradients on the trainer's second stream (and the alternating gradient buffers): repeated
    forward/backward passes of the full-size step (640x480, 11 frames) from the same weights give the same gradients up to the order
    of the float atomics (~3e-7 of the largest gradient); a stale read of an overwritten buffer would show as a jump."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "soak_train_overlap.py"), "8"], cwd=root, capture_output=True,
                       text=True, timeout=900, env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-800:])
    worst = float(r.stdout.strip().splitlines()[-1].split("first pass")[1].split()[0])
    assert worst < 1e-5, r.stdout[-400:]


def test_dgrad_rejects_an_h2_gate_on_fewer_than_64_channels(lib_built):
    """dx with 32 channels goes through the 32-column fp32 tile, which reads the gate as fp32: with H2 cells the result was garbage
    (found by scripts/fuzz_backward_layers.py; the network has no such layer).  DGP_ERR_INVALID now; the fp32 gate works."""
    from deepgraphpose_amd import _lib, engine
    g = torch.Generator(device="cuda").manual_seed(5)
    dy = torch.randn((2, 7, 9, 64), generator=g, device="cuda") * 1e-3
    w = torch.randn((1, 1, 32, 64), generator=g, device="cuda") / 6.0
    mask = torch.relu(torch.randn((2, 7, 9, 32), generator=g, device="cuda"))
    with pytest.raises(_lib.DgpError, match="Cin >= 64"):
        engine.conv2d_dgrad(dy, w, (7, 9), mask=mask, ranged=True, mask_h2=True)
    dx = engine.conv2d_dgrad(dy, w, (7, 9), mask=mask, ranged=True, mask_h2=False)
    ref = torch.where(mask > 0, _dgrad_ref(dy.double(), w.double(), 7, 9, 1, 1, 0, 0), torch.zeros((2, 7, 9, 32), dtype=torch.float64, device="cuda"))
    assert float((dx.double() - ref).abs().max() / ref.abs().max()) < REL_TOL


@pytest.mark.parametrize("depth,nj", [(50, 4), (101, 20)])
def test_full_size_config4_step_on_the_16_bit_tier(lib_built, depth, nj):
    """BASELINE configs[3] at FULL size on the tier it names (16-bit): 640 x 480, 11 frames (1 labeled + 10 unlabeled), gm2 = 1, gm3 = 3 --
    ResNet-50 with 4 keypoints on a chain skeleton, and ResNet-101 with 20 keypoints (configs[4]'s network) -- against the autograd oracle
    evaluated on the host cores at run time (fp32: its own distance from float64 is ~1e-3 of the 16-bit tier's band).  The SAME bounds as
    the small-shape gradient tests of tests/test_train_gpu.py: loss within 2e-3 relative, global gradient cosine >= 0.999, relative L2
    error <= 5 %, every tensor's cosine >= 0.98 -- at the real grid sizes (104-208-tile H1 launches, 384-workgroup weight-gradient
    grids, the fused stem kernels on 120 x 160 maps).  Pass 1 is the shape's parity pass and meets the parity bounds."""
    import ctypes
    from test_train_gpu import _make_loss_case, _oracle_grads, _grad_agreement
    from deepgraphpose_amd.synthetic import make_weights, make_frames
    from deepgraphpose_amd.train import Trainer
    from deepgraphpose_amd.loss import DGPHyper
    torch.set_num_threads(min(32, torch.get_num_threads() * 2 if torch.get_num_threads() < 16 else 32))
    nt, hw = 11, (480, 640)
    nl = 3 if nj == 4 else 19
    rng = np.random.default_rng(44 + depth)
    batch, _ = _make_loss_case(rng, nt, 60, 80, nj, 1, 0.0, 2)
    S0 = np.zeros((nl, nj))
    for l in range(nl):
        S0[l, l], S0[l, l + 1] = 1, -1
    wts = make_weights(depth, nj, True, seed=4, head_std=0.05)
    frames = make_frames(nt, hw[0], hw[1], nj, seed=4)
    ws, ws_max = rng.uniform(5, 20, nl), rng.uniform(10, 40, nl)
    hy = DGPHyper(gm2=1, gm3=3)
    n_tot, n_vis = 1000.0, 50.0
    tr = Trainer(depth, nj, hw[0], hw[1], max_frames=nt, tier="f16")
    tr.load_weights(wts)
    ft = torch.from_numpy(frames).cuda()
    was, failed = ctypes.c_int32(), ctypes.c_int32()
    l0 = tr.forward_backward(ft, batch, hy, S0, ws, ws_max, n_tot, n_vis)            # first pass of the shape: parity path
    tr.lib.dgp_trainer_fast_status(tr._t, was, failed)
    assert was.value == 0
    g0 = tr.get_grads()
    l1 = tr.forward_backward(ft, batch, hy, S0, ws, ws_max, n_tot, n_vis)            # the 16-bit pass
    tr.lib.dgp_trainer_fast_status(tr._t, was, failed)
    assert was.value == 1 and failed.value == 0 and tr.fast_passes == 1 and tr.fast_redos == 0
    g1 = tr.get_grads()
    P, L = _oracle_grads(wts, frames, batch, S0, ws, ws_max, hy, n_tot, n_vis, depth=depth, dtype=torch.float32)
    ref_loss = float(L["total_loss"].detach())
    cos0, rel0, _ = _grad_agreement(g0, P)
    cos1, rel1, per = _grad_agreement(g1, P)
    worst = sorted(per.items(), key=lambda kv: kv[1][0])[:3]
    print("full size, ResNet-%d nj %d: loss parity %.6f / f16 %.6f (oracle %.6f) | parity pass cosine %.6f rel %.2e | 16-bit pass cosine %.6f rel %.4f | worst %s"
          % (depth, nj, l0["total_loss"], l1["total_loss"], ref_loss, cos0, rel0, cos1, rel1, worst))
    assert abs(l0["total_loss"] - ref_loss) < 2e-4 * max(1, abs(ref_loss)) and cos0 > 0.99999 and rel0 < 3e-3
    assert abs(l1["total_loss"] - ref_loss) < 2e-3 * max(1, abs(ref_loss))
    assert cos1 >= 0.999 and rel1 <= 0.05
    assert min(v[0] for v in per.values()) >= 0.98, worst
    assert all(np.isfinite(v).all() for v in g1.values())
