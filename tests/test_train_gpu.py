"""GPU parity tests of the training-step kernels vs the torch-autograd oracle (oracle/dgp_train_oracle.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make_loss_case(rng, nt, H, W, nj, n_vis_frames, nan_frac, nl):
    from deepgraphpose_amd import dataset as D
    vis_frames = np.sort(rng.choice(nt, n_vis_frames, replace=False))
    hid_frames = np.setdiff1d(np.arange(nt), vis_frames)
    jl = np.stack([rng.uniform(1, H - 2, (n_vis_frames, nj)), rng.uniform(1, W - 2, (n_vis_frames, nj))], -1)
    jl[rng.random((n_vis_frames, nj)) < nan_frac] = np.nan
    vm, hm, vt = D.gen_idx_chunk(vis_frames, hid_frames, jl)
    lt, lm = D.coord2map(jl, H, W, nj, 8) if n_vis_frames else (np.zeros((0, H, W, 2 * nj)), np.zeros((0, H, W, 2 * nj)))
    lmap, lmask = np.zeros((nt, H, W, 2 * nj)), np.zeros((nt, H, W, 2 * nj))
    if n_vis_frames:
        lmap[vis_frames], lmask[vis_frames] = lt, lm
    S0 = np.zeros((nl, nj))
    for l in range(nl):
        a, b = rng.choice(nj, 2, replace=False)
        S0[l, a], S0[l, b] = 1, -1
    batch = dict(targets=jl, locref_map=lmap, locref_mask=lmask, visible_marker=vm, hidden_marker=hm,
                 visible_marker_in_targets=vt, nt=nt)
    return batch, S0


@pytest.mark.parametrize("gm2,gm3", [(0, 0), (1, 3), (2, 3), (1, 0)])
@pytest.mark.parametrize("shape", [(5, 12, 16, 3, 2), (11, 60, 80, 4, 1), (3, 9, 7, 2, 0), (2, 10, 10, 3, 2)])
def test_loss_forward_backward_matches_autograd(lib_built, gm2, gm3, shape):
    from deepgraphpose_amd.loss import dgp_loss_fwd_bwd, DGPHyper
    from oracle import dgp_train_oracle as T
    nt, H, W, nj, nvf = shape
    rng = np.random.default_rng(nt * 100 + H + gm2 * 7 + gm3)
    nl = 0 if nj < 2 else 2
    if shape == (2, 10, 10, 3, 2):           # all frames visible, some NaN joints -> hidden markers in visible frames
        batch, S0 = _make_loss_case(rng, nt, H, W, nj, nvf, 0.4, nl)
    else:
        batch, S0 = _make_loss_case(rng, nt, H, W, nj, nvf, 0.2, nl)
    pred = (rng.standard_normal((nt, H, W, nj)) * 2).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for n in range(nt):
        for j in range(nj):
            cy, cx = rng.uniform(0, H - 1), rng.uniform(0, W - 1)
            pred[n, :, :, j] += 6 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / 6.0)
    loc = rng.standard_normal((nt, H, W, 2 * nj)).astype(np.float32)
    hy = DGPHyper(gm2=gm2, gm3=gm3)
    ws = rng.uniform(5, 20, S0.shape[0])
    ws_max = rng.uniform(10, 40, S0.shape[0])
    n_tot, n_vis_tot = 500.0, 37.0
    cfg = dict(nj=nj, S0=S0, ws=ws, ws_max=ws_max, stride=8.0, gamma=hy.gamma, gauss_len=hy.gauss_len,
               lengthscale=hy.lengthscale, gm2=gm2, gm3=gm3, wn_visible=hy.wn_visible, wn_hidden=hy.wn_hidden,
               locref_loss_weight=hy.locref_loss_weight, locref_huber_loss=True, n_frames_total=n_tot,
               n_visible_frames_total=n_vis_tot)
    # oracle in float64 (truth) -- the kernels are fp32 with fp64 reductions
    pt = torch.tensor(pred, dtype=torch.float64, requires_grad=True)
    lt = torch.tensor(loc, dtype=torch.float64, requires_grad=True)
    L = T.dgp_loss(pt, lt, batch, cfg)
    L["total_loss"].backward()
    losses, dpred, dloc, mu = dgp_loss_fwd_bwd(torch.from_numpy(pred).cuda(), torch.from_numpy(loc).cuda(), batch, hy,
                                               S0, ws, ws_max, n_tot, n_vis_tot)
    for k in ("visible_loss_pred", "hidden_loss_pred", "visible_loss_locref", "total_loss", "total_loss_visible"):
        assert abs(losses[k] - float(L[k])) <= 2e-5 * max(1.0, abs(float(L[k]))), (k, losses[k], float(L[k]))
    if S0.shape[0]:
        assert abs(losses["ws_loss"] - float(L["ws_loss"])) <= 2e-5 * max(1.0, abs(float(L["ws_loss"])))
    np.testing.assert_allclose(mu.cpu().numpy(), L["_mu"].detach().numpy(), atol=2e-5)
    gp, gl = pt.grad.numpy(), lt.grad.numpy() if lt.grad is not None else np.zeros_like(loc)
    scale = np.abs(gp).max() + 1e-12
    assert np.abs(dpred.cpu().numpy() - gp).max() <= 2e-4 * scale + 1e-9
    assert np.abs(dloc.cpu().numpy() - gl).max() <= 2e-5 * (np.abs(gl).max() + 1e-12) + 1e-10


def _loss_case_with_peaks(rng, nt, H, W, nj, nvf, nl, gm2, gm3):
    from deepgraphpose_amd.loss import DGPHyper
    batch, S0 = _make_loss_case(rng, nt, H, W, nj, nvf, 0.2, nl)
    pred = (rng.standard_normal((nt, H, W, nj)) * 2).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for n in range(nt):
        for j in range(nj):
            cy, cx = rng.uniform(0, H - 1), rng.uniform(0, W - 1)
            pred[n, :, :, j] += 6 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / 6.0)
    loc = rng.standard_normal((nt, H, W, 2 * nj)).astype(np.float32)
    hy = DGPHyper(gm2=gm2, gm3=gm3)
    ws, ws_max = rng.uniform(5, 20, S0.shape[0]), rng.uniform(10, 40, S0.shape[0])
    cfg = dict(nj=nj, S0=S0, ws=ws, ws_max=ws_max, stride=8.0, gamma=hy.gamma, gauss_len=hy.gauss_len,
               lengthscale=hy.lengthscale, gm2=gm2, gm3=gm3, wn_visible=hy.wn_visible, wn_hidden=hy.wn_hidden,
               locref_loss_weight=hy.locref_loss_weight, locref_huber_loss=True, n_frames_total=500.0, n_visible_frames_total=37.0)
    return batch, S0, pred, loc, hy, ws, ws_max, cfg


@pytest.mark.parametrize("shape,gm2,gm3", [((3, 136, 240, 4, 1), 1, 3), ((2, 150, 200, 2, 1), 0, 0), ((2, 120, 170, 3, 1), 2, 3)])
def test_loss_on_maps_beyond_the_lds_limit_matches_autograd(lib_built, shape, gm2, gm3):
    """The reference's loss placeholders are [None, None, None, nj] (DGP/models/fitdgp.py:1130-1142): any scoremap size.  Beyond 19 200
    cells (frames above ~960 x 1280; 136 x 240 is a 1080p frame) the two per-marker maps no longer fit the LDS and loss_ce_backward
    streams: Gaussian target, sigmoid and softmax values recomputed where they are read.  Same bounds as the LDS-resident maps, against
    the float64 autograd oracle: CE terms (gm2 / gm3), Gaussian targets through the hidden markers' soft-argmax, locref Huber, clique."""
    from deepgraphpose_amd.loss import dgp_loss_fwd_bwd
    from oracle import dgp_train_oracle as T
    nt, H, W, nj, nvf = shape
    assert 2 * H * W * 4 > 150 * 1024
    rng = np.random.default_rng(H * 7 + W + gm2)
    batch, S0, pred, loc, hy, ws, ws_max, cfg = _loss_case_with_peaks(rng, nt, H, W, nj, nvf, 2 if nj > 2 else (1 if nj == 2 else 0), gm2, gm3)
    pt = torch.tensor(pred, dtype=torch.float64, requires_grad=True)
    lt = torch.tensor(loc, dtype=torch.float64, requires_grad=True)
    L = T.dgp_loss(pt, lt, batch, cfg)
    L["total_loss"].backward()
    losses, dpred, dloc, mu = dgp_loss_fwd_bwd(torch.from_numpy(pred).cuda(), torch.from_numpy(loc).cuda(), batch, hy, S0, ws, ws_max, 500.0, 37.0)
    for k in ("visible_loss_pred", "hidden_loss_pred", "visible_loss_locref", "total_loss", "total_loss_visible"):
        assert abs(losses[k] - float(L[k])) <= 2e-5 * max(1.0, abs(float(L[k]))), (k, losses[k], float(L[k]))
    np.testing.assert_allclose(mu.cpu().numpy(), L["_mu"].detach().numpy(), atol=2e-5)
    gp, gl = pt.grad.numpy(), lt.grad.numpy()
    assert np.abs(dpred.cpu().numpy() - gp).max() <= 2e-4 * (np.abs(gp).max() + 1e-12) + 1e-9
    assert np.abs(dloc.cpu().numpy() - gl).max() <= 2e-5 * (np.abs(gl).max() + 1e-12) + 1e-10


@pytest.mark.parametrize("gm2,gm3", [(1, 3), (0, 0), (2, 0)])
def test_streaming_loss_kernel_is_bit_identical_to_the_lds_variant(lib_built, gm2, gm3):
    """Both instances of loss_ce_backward on the SAME maps (DGP_LOSS_STREAM=1 forces the streaming one): losses, gradients and mu equal bit
    for bit -- the recomputed Gaussian / sigmoid / softmax values come from the same out-of-line functions, so the `== max` tests of the
    reduce_max gradients see the same bits in both (a result cannot change at the 19 200-cell boundary)."""
    import os
    from deepgraphpose_amd.loss import dgp_loss_fwd_bwd
    rng = np.random.default_rng(11 + gm2)
    batch, S0, pred, loc, hy, ws, ws_max, _ = _loss_case_with_peaks(rng, 4, 60, 80, 4, 1, 2, gm2, gm3)
    p, l = torch.from_numpy(pred).cuda(), torch.from_numpy(loc).cuda()

    def run():
        losses, dpred, dloc, mu = dgp_loss_fwd_bwd(p, l, batch, hy, S0, ws, ws_max, 500.0, 37.0)
        torch.cuda.synchronize()
        return losses, dpred.clone(), dloc.clone(), mu.clone()
    a = run()
    os.environ["DGP_LOSS_STREAM"] = "1"
    try:
        b = run()
    finally:
        del os.environ["DGP_LOSS_STREAM"]
    # (the loss VALUES are sums of per-marker terms added with float atomics in whatever order the workgroups finish: compared to 1e-6)
    for k in a[0]:
        assert abs(a[0][k] - b[0][k]) <= 1e-6 * max(1.0, abs(a[0][k])), k
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])


# ---------------------------------------------------------------------------------------------------------------
def _train_case(seed, hw=(64, 96), nt=3, nj=3, nvf=1, depth=50):
    from deepgraphpose_amd.synthetic import make_weights, make_frames
    from deepgraphpose_amd.arch import scoremap_hw
    rng = np.random.default_rng(seed)
    H, W = scoremap_hw(*hw)
    nl = 2 if nj >= 3 else (1 if nj == 2 else 0)
    batch, S0 = _make_loss_case(rng, nt, H, W, nj, nvf, 0.0, nl)
    wts = make_weights(depth, nj, True, seed=seed, head_std=0.05)
    frames = make_frames(nt, hw[0], hw[1], nj, seed=seed)
    ws, ws_max = rng.uniform(5, 20, nl), rng.uniform(10, 40, nl)
    return batch, S0, wts, frames, ws, ws_max


def _oracle_grads(wts, frames, batch, S0, ws, ws_max, hy, n_tot, n_vis, depth=50, dtype=torch.float32):
    from oracle import dgp_train_oracle as T
    P = T.make_params(wts, dtype)
    pred, loc = T.network(frames, P, depth, dtype)
    nj = pred.shape[-1]
    cfg = dict(nj=nj, S0=S0, ws=ws, ws_max=ws_max, stride=8.0, gamma=hy.gamma, gauss_len=hy.gauss_len,
               lengthscale=hy.lengthscale, gm2=hy.gm2, gm3=hy.gm3, wn_visible=hy.wn_visible, wn_hidden=hy.wn_hidden,
               locref_loss_weight=hy.locref_loss_weight, locref_huber_loss=True, n_frames_total=n_tot,
               n_visible_frames_total=n_vis)
    L = T.dgp_loss(pred, loc, batch, cfg)
    L["total_loss"].backward()
    return P, L


def test_full_backward_matches_autograd(lib_built):
    """Gradients of every trainable tensor (53 convs x {W, gamma, beta} + 2 heads x {W, b}) vs torch autograd."""
    from deepgraphpose_amd.train import Trainer
    from deepgraphpose_amd.loss import DGPHyper
    batch, S0, wts, frames, ws, ws_max = _train_case(3)
    hy = DGPHyper(gm2=1, gm3=3)
    n_tot, n_vis = 300.0, 25.0
    P, L = _oracle_grads(wts, frames, batch, S0, ws, ws_max, hy, n_tot, n_vis, dtype=torch.float64)
    tr = Trainer(50, 3, 64, 96, max_frames=3)
    tr.load_weights(wts)
    losses = tr.forward_backward(torch.from_numpy(frames).cuda(), batch, hy, S0, ws, ws_max, n_tot, n_vis)
    assert abs(losses["total_loss"] - float(L["total_loss"].detach())) < 1e-4 * max(1, abs(float(L["total_loss"].detach())))
    g = tr.get_grads()
    # A ReLU whose pre-activation is ~1e-7 can come out on the other side of zero in the fp32 HIP forward than in
    # the fp64 oracle (seen: one element of 73 728 in block3/unit_6, activation 7e-7 vs 0) -- the gate then differs
    # for that single element and every layer upstream of it inherits an O(1e-3) relative L2 error on these tiny
    # 4x6 feature maps.  So: layers downstream of any gate flip (block4 + heads) must agree to fp32 round-off,
    # all others to 1e-2 in relative L2, and the global gradient norm to 1e-3.
    rel, tot_ref, tot_err = {}, 0.0, 0.0
    for k, t in P.items():
        if not t.requires_grad:
            continue
        ref = t.grad.numpy()
        d = g[k].reshape(ref.shape) - ref
        rel[k] = np.linalg.norm(d.ravel()) / (np.linalg.norm(ref.ravel()) + 1e-30)
        tot_ref += float((ref ** 2).sum())
        tot_err += float((d ** 2).sum())
    strict = {k: v for k, v in rel.items() if "block4" in k or k.startswith("pose/")}
    assert len(strict) == 3 * 10 + 4 and max(strict.values()) < 2e-5, sorted(strict.items(), key=lambda kv: -kv[1])[:4]
    assert max(rel.values()) < 1e-2, sorted(rel.items(), key=lambda kv: -kv[1])[:4]
    assert np.sqrt(tot_err / tot_ref) < 3e-3


def test_product_build_refuses_the_opt_in_fast_pass(lib_built):
    """The fast pass measured no faster than the plain pass and lives in -DDGP_TUNING builds only: the product library says so instead
    of silently running something else."""
    from deepgraphpose_amd import _lib
    from deepgraphpose_amd.train import Trainer
    lib = _lib.load()
    if lib.dgp_tuning_build():
        pytest.skip("tuning build: the fast pass is available")
    tr = Trainer(50, 3, 64, 96, max_frames=3)
    assert lib.dgp_trainer_fast_mode(tr._t, 0) == 0
    assert lib.dgp_trainer_fast_mode(tr._t, 1) == -1 and b"DGP_TUNING" in lib.dgp_last_error()       # DGP_ERR_INVALID


def test_fast_pass_matches_autograd_and_recovers_from_a_failed_prediction(lib_built, tuning_build, monkeypatch):
    """DGP_TRAIN_H2=1: from the second pass on, blocks 2-4 keep their activations as H2 tensors with scales predicted from the previous
    pass (dgp_trainer_fast_mode).  (a) the fast pass meets the plain pass's tolerances against the fp64 oracle; (b) when the ranges jump
    by more than the predicted scales cover (here: new weights 64 x larger in the stem, smuggled in behind the trainer's back), the
    step reports it and is repeated as a plain pass before anything is returned."""
    from deepgraphpose_amd.train import Trainer
    from deepgraphpose_amd.loss import DGPHyper
    monkeypatch.setenv("DGP_TRAIN_H2", "1")
    batch, S0, wts, frames, ws, ws_max = _train_case(3)
    hy = DGPHyper(gm2=1, gm3=3)
    n_tot, n_vis = 300.0, 25.0
    tr = Trainer(50, 3, 64, 96, max_frames=3)
    tr.load_weights(wts)
    ft = torch.from_numpy(frames).cuda()

    def check(wts_now):
        P, L = _oracle_grads(wts_now, frames, batch, S0, ws, ws_max, hy, n_tot, n_vis, dtype=torch.float64)
        g = tr.get_grads()
        rel, tot_ref, tot_err = {}, 0.0, 0.0
        for k, t in P.items():
            if not t.requires_grad:
                continue
            ref = t.grad.numpy()
            d = g[k].reshape(ref.shape) - ref
            rel[k] = np.linalg.norm(d.ravel()) / (np.linalg.norm(ref.ravel()) + 1e-30)
            tot_ref += float((ref ** 2).sum())
            tot_err += float((d ** 2).sum())
        strict = {k: v for k, v in rel.items() if "block4" in k or k.startswith("pose/")}
        assert max(strict.values()) < 2e-5, sorted(strict.items(), key=lambda kv: -kv[1])[:4]
        assert max(rel.values()) < 1e-2, sorted(rel.items(), key=lambda kv: -kv[1])[:4]
        assert np.sqrt(tot_err / tot_ref) < 3e-3
        return L

    tr.forward_backward(ft, batch, hy, S0, ws, ws_max, n_tot, n_vis)                 # plain pass: leaves the ranges behind
    was, failed = __import__("ctypes").c_int32(), __import__("ctypes").c_int32()
    tr.lib.dgp_trainer_fast_status(tr._t, was, failed)
    assert was.value == 0
    losses = tr.forward_backward(ft, batch, hy, S0, ws, ws_max, n_tot, n_vis)        # fast pass
    tr.lib.dgp_trainer_fast_status(tr._t, was, failed)
    assert was.value == 1 and failed.value == 0 and getattr(tr, "fast_redos", 0) == 0
    L = check(wts)
    assert abs(losses["total_loss"] - float(L["total_loss"].detach())) < 1e-4 * max(1, abs(float(L["total_loss"].detach())))
    # (b) a 64-fold jump of every activation range: upload the new stem scale without telling the Python wrapper
    w2 = dict(wts)
    w2["resnet_v1_50/conv1/BatchNorm/gamma"] = wts["resnet_v1_50/conv1/BatchNorm/gamma"] * 64.0
    w2["resnet_v1_50/conv1/BatchNorm/beta"] = wts["resnet_v1_50/conv1/BatchNorm/beta"] * 64.0
    key = tr._fast_key
    tr.load_weights(w2)
    tr._fast_key = key                                                                # pretend nothing happened: the next pass goes fast
    losses = tr.forward_backward(ft, batch, hy, S0, ws, ws_max, n_tot, n_vis)
    assert tr.fast_redos == 1                                                         # ... fails its range check and is repeated
    tr.lib.dgp_trainer_fast_status(tr._t, was, failed)
    assert was.value == 0
    L = check(w2)
    assert abs(losses["total_loss"] - float(L["total_loss"].detach())) < 1e-4 * max(1, abs(float(L["total_loss"].detach())))
    tr.forward_backward(ft, batch, hy, S0, ws, ws_max, n_tot, n_vis)                 # and the pass after that is fast again
    tr.lib.dgp_trainer_fast_status(tr._t, was, failed)
    assert was.value == 1 and failed.value == 0 and tr.fast_redos == 1
    check(w2)


def test_flat_gradient_view_matches_named_gradients(lib_built):
    """N4 plumbing: the flat device view that the RCCL all-reduce averages is the same memory get_grads() downloads."""
    from deepgraphpose_amd.train import Trainer
    from deepgraphpose_amd.loss import DGPHyper
    batch, S0, wts, frames, ws, ws_max = _train_case(7)
    tr = Trainer(50, 3, 64, 96, max_frames=3)
    tr.load_weights(wts)
    tr.forward_backward(torch.from_numpy(frames).cuda(), batch, DGPHyper(gm2=0, gm3=0), S0, ws, ws_max, 300.0, 25.0)
    flat = tr.grads_tensor().cpu().numpy()
    named = tr.get_grads()
    assert flat.size == tr.n_trainable
    for k, (off, size, st) in tr.table.items():
        if not st:
            np.testing.assert_array_equal(flat[off:off + size], named[k].ravel())
    tr.allreduce_gradients()                                    # world size 1: a no-op that must not touch the buffer
    np.testing.assert_array_equal(tr.grads_tensor().cpu().numpy(), flat)


def test_two_optimizer_steps_match_oracle(lib_built):
    from deepgraphpose_amd.train import Trainer
    from deepgraphpose_amd.loss import DGPHyper
    from oracle import dgp_train_oracle as T
    batch, S0, wts, frames, ws, ws_max = _train_case(5, nvf=2)
    hy = DGPHyper(gm2=0, gm3=0, lr=0.005)
    n_tot, n_vis = 300.0, 25.0
    tr = Trainer(50, 3, 64, 96, max_frames=3)
    tr.load_weights(wts)
    P = T.make_params(wts, torch.float32)
    V = {}
    ft = torch.from_numpy(frames).cuda()
    for it in range(2):
        losses = tr.step(ft, batch, hy, S0, ws, ws_max, n_tot, n_vis)
        pred, loc = T.network(frames, P, 50, torch.float32)
        cfg = dict(nj=3, S0=S0, ws=ws, ws_max=ws_max, stride=8.0, gamma=1.0, gauss_len=1, lengthscale=1.0, gm2=0, gm3=0,
                   wn_visible=5.0, wn_hidden=3.0, locref_loss_weight=0.05, locref_huber_loss=True, n_frames_total=n_tot,
                   n_visible_frames_total=n_vis)
        L = T.dgp_loss(pred, loc, batch, cfg)
        L["total_loss"].backward()
        gn = T.momentum_step(P, V, hy.lr)
        assert abs(losses["total_loss"] - float(L["total_loss"].detach())) < 2e-4 * max(1.0, abs(float(L["total_loss"].detach())))
        assert abs(losses["grad_norm"] - gn) < 2e-3 * gn
    w = tr.get_weights()
    for k, t in P.items():
        ref = t.detach().numpy()
        assert np.abs(w[k].reshape(ref.shape) - ref).max() <= 1e-4 * (np.abs(ref).max() + 1e-6) + 1e-6, k


def test_fit_dgp_drivers_end_to_end(lib_built, tmp_path):
    """fit_dgp_labeledonly (step 1) -> fit_dgp (step 2, gm2=1 gm3=3 like the demo) -> estimate_pose on a synthetic DLC
    project: runs, writes the step snapshots, losses finite and the labeled-only loss decreases on its frame."""
    import os, random
    from _project import make_project
    from deepgraphpose_amd.models.fitdgp import fit_dgp, fit_dgp_labeledonly
    from deepgraphpose_amd.models.fitdgp_util import get_snapshot_path
    from deepgraphpose_amd.models.eval import estimate_pose
    from deepgraphpose_amd import weights_io
    proj, frames, wts = make_project(tmp_path)
    np.random.seed(0); random.seed(0)
    fit_dgp_labeledonly("snapshot-step0-final--0", proj, shuffle=1, step=1, maxiters=4, displayiters=1, aug=False)
    snap1, cfg_path = get_snapshot_path("snapshot-step1-final--0", proj, shuffle=1)
    assert os.path.isfile(snap1 + ".index") and os.path.isfile(snap1 + ".data-00000-of-00001")
    w1 = weights_io.load_weights(snap1)
    assert np.abs(w1["pose/part_pred/block4/weights"] - wts["pose/part_pred/block4/weights"]).max() > 0
    fit_dgp("snapshot-step1-final--0", proj, batch_size=4, shuffle=1, step=2, maxiters=3, displayiters=1, gm2=1, gm3=3,
            aug=False, n_max_frames=30, ns=3)
    snap2, _ = get_snapshot_path("snapshot-step2-final--0", proj, shuffle=1)
    assert os.path.isfile(snap2 + ".index") and os.path.isfile(snap2 + ".data-00000-of-00001")
    assert weights_io.latest_checkpoint(os.path.dirname(snap2)) == snap2          # the Saver's `checkpoint` state file
    w2 = weights_io.load_weights(snap2)
    assert all(np.isfinite(v).all() for v in w2.values())
    # second call is a no-op (skip-if-exists guard, fitdgp.py:656-660)
    assert fit_dgp("snapshot-step1-final--0", proj, batch_size=4, maxiters=3) is None
    labels = estimate_pose(str(cfg_path), snap2, os.path.join(proj, "videos", "clip.npy"), os.path.join(proj, "videos_pred"),
                           shuffle=1, batch_size=8)
    assert labels["x"].shape == (40, 3) and np.isfinite(labels["x"]).all()
    assert os.path.isfile(os.path.join(proj, "videos_pred", "clip_labeled.csv"))
    # N1 (a): the trajectory estimate_pose computed FROM THE TF BUNDLE fit_dgp wrote equals the oracle's on the tensors an independent
    # reader (tests/_kat_ckpt.py: own table walk, own CRC) finds in that bundle -- coordinates within 1e-3 px, window indices bit-exact
    import _kat_ckpt as kat
    from oracle import dgp_oracle as O
    disk = kat.read_bundle(snap2)
    assert sorted(disk) == sorted(w2) and all(np.array_equal(disk[k], w2[k]) for k in w2)
    ref = O.infer(frames, disk, 50, 8.0, 1.0, 1)
    assert np.abs(labels["x"] - ref["x"]).max() < 1e-3 and np.abs(labels["y"] - ref["y"]).max() < 1e-3
    assert np.abs(labels["likelihoods"] - ref["likelihoods"]).max() < 1e-5
    from deepgraphpose_amd import engine
    import torch
    net = engine.DGPNet(50, 3, 64, 96, max_batch=8, device=0)
    net.load_weights(weights_io.load_weights(snap2))
    _, _, idx = net.infer(torch.from_numpy(frames[:8]).cuda(), 1.0, 1)
    assert np.array_equal(idx.cpu().numpy(), ref["idx"][:8])
    # evaluate_dgp (eval.py:656): RMSE table over the labeled images, both read-out paths
    from deepgraphpose_amd.models.eval import evaluate_dgp
    for lr in (True, False):
        rmse = evaluate_dgp(str(cfg_path), snap2, shuffle=1, loc_ref=lr)
        assert rmse.shape == (4, 3) and np.isfinite(rmse.values[~np.isnan(rmse.values)]).all()
        assert np.isnan(rmse.values).sum() == 1          # the one unlabeled joint


def test_temporal_clique_matches_oracle(lib_built):
    """B8 (wt > 0): temporal graph-smoothness term on top of the other terms, flow-weighted, vs the autograd oracle -- including
    the gradient through the flow weights (fitdgp.py:1091-1113: the boxes of tf.image.crop_and_resize are functions of the targets)."""
    from deepgraphpose_amd.loss import dgp_loss_fwd_bwd, DGPHyper
    from oracle import dgp_train_oracle as T
    nt, H, W, nj, nvf = 6, 12, 16, 3, 2
    rng = np.random.default_rng(42)
    batch, S0 = _make_loss_case(rng, nt, H, W, nj, nvf, 0.0, 2)
    Hin, Win = 8 * H, 8 * W
    yy, xx = np.mgrid[0:Hin, 0:Win]
    vf = np.stack([np.abs(np.sin(xx / 9.0 + t) * np.cos(yy / 7.0)) * rng.uniform(0.2, 3.0) for t in range(nt - 1)]).astype(np.float32)
    batch["vector_field"] = vf
    batch["wt_batch_mask"] = np.array([1, 1, 0, 1, 1], dtype=np.float32)
    pred = (rng.standard_normal((nt, H, W, nj)) * 2).astype(np.float32)
    for n in range(nt):
        for j in range(nj):
            cy, cx = rng.uniform(1, H - 2), rng.uniform(1, W - 2)
            pred[n, :, :, j] += 7 * np.exp(-((np.mgrid[0:H, 0:W][0] - cy) ** 2 + (np.mgrid[0:H, 0:W][1] - cx) ** 2) / 4.0)
    loc = rng.standard_normal((nt, H, W, 2 * nj)).astype(np.float32)
    for wt_max in (0.0, 6.0):
        hy = DGPHyper(gm2=1, gm3=3, wt=50.0, wt_max=wt_max)
        ws, ws_max = rng.uniform(5, 20, 2), rng.uniform(10, 40, 2)
        cfg = dict(nj=nj, S0=S0, ws=ws, ws_max=ws_max, stride=8.0, gamma=1.0, gauss_len=1, lengthscale=1.0, gm2=1, gm3=3,
                   wn_visible=5.0, wn_hidden=3.0, locref_loss_weight=0.05, locref_huber_loss=True, n_frames_total=500.0,
                   n_visible_frames_total=37.0, wt=50.0, wt_max=wt_max)
        pt = torch.tensor(pred, dtype=torch.float64, requires_grad=True)
        lt = torch.tensor(loc, dtype=torch.float64, requires_grad=True)
        L = T.dgp_loss(pt, lt, batch, cfg)
        L["total_loss"].backward()
        losses, dpred, dloc, mu = dgp_loss_fwd_bwd(torch.from_numpy(pred).cuda(), torch.from_numpy(loc).cuda(), batch, hy, S0, ws,
                                                   ws_max, 500.0, 37.0)
        assert float(L["wt_loss"].detach()) > 0
        assert abs(losses["wt_loss"] - float(L["wt_loss"].detach())) < 1e-4 * float(L["wt_loss"].detach())
        assert abs(losses["total_loss"] - float(L["total_loss"].detach())) < 1e-4 * abs(float(L["total_loss"].detach()))
        gp = pt.grad.numpy()
        assert np.abs(dpred.cpu().numpy() - gp).max() <= 3e-4 * np.abs(gp).max()
        # the gradient THROUGH the flow weights (the crop boxes follow the differentiable targets; TF's crop_and_resize has a box
        # gradient) is part of it: with the weight held constant the oracle's gradient is measurably different
        pt2 = torch.tensor(pred, dtype=torch.float64, requires_grad=True)
        lt2 = torch.tensor(loc, dtype=torch.float64, requires_grad=True)
        L2 = T.dgp_loss(pt2, lt2, batch, dict(cfg, wt_weight_grad=False))
        L2["total_loss"].backward()
        assert abs(float(L2["wt_loss"].detach()) - float(L["wt_loss"].detach())) < 1e-12
        gap = np.abs(pt2.grad.numpy() - gp).max()
        assert gap > 30 * np.abs(dpred.cpu().numpy() - gp).max(), gap


def _dlc_targets(rng, H, W, nj, scale=1.0):
    from deepgraphpose_amd.dataset import compute_target_part_scoremap
    oh, ow = 2 * -(-H // 16), 2 * -(-W // 16)
    ids = [np.sort(rng.choice(nj, nj - 1, replace=False))]
    coords = [np.stack([rng.uniform(4, W - 4, nj - 1), rng.uniform(4, H - 4, nj - 1)], 1)]
    sc, lmap, lmask = compute_target_part_scoremap(ids, coords, (oh, ow), nj, 8, scale=scale)
    return sc[None].astype(np.float32), lmap[None].astype(np.float32), lmask[None].astype(np.float32)


@pytest.mark.parametrize("weighted", [False, True])
def test_dlc_loss_kernel_matches_autograd(lib_built, weighted):
    """N2: sigmoid-CE on binary disks + masked Huber (pose_net.train) forward and d/d heads vs fp64 autograd."""
    import ctypes as C
    from deepgraphpose_amd import _lib
    from deepgraphpose_amd.engine import _ptr
    from oracle import dgp_train_oracle as T
    rng = np.random.default_rng(5)
    nt, H, W, nj = 2, 23, 31, 3
    pred = (rng.standard_normal((nt, H, W, nj)) * 3).astype(np.float32)
    loc = (rng.standard_normal((nt, H, W, 2 * nj)) * 1.5).astype(np.float32)
    pt = (rng.random((nt, H, W, nj)) < 0.1).astype(np.float32)
    lm = np.repeat(pt, 2, axis=3)
    lt = rng.standard_normal((nt, H, W, 2 * nj)).astype(np.float32) * lm
    pw = None
    if weighted:
        pw = np.ones_like(pt); pw[..., 1] = 0.0
    tp = torch.tensor(pred, dtype=torch.float64, requires_grad=True)
    tl = torch.tensor(loc, dtype=torch.float64, requires_grad=True)
    L = T.dlc_loss(tp, tl, torch.tensor(pt, dtype=torch.float64), torch.tensor(lt, dtype=torch.float64),
                   torch.tensor(lm, dtype=torch.float64), None if pw is None else torch.tensor(pw, dtype=torch.float64), 0.05)
    L["total_loss"].backward()
    lib = _lib.load()
    d = lambda a: torch.from_numpy(a).cuda()
    dp, dl = torch.empty(pred.shape, device="cuda"), torch.empty(loc.shape, device="cuda")
    losses, scratch = torch.empty(4, device="cuda"), torch.empty(4, dtype=torch.float64, device="cuda")
    args = [d(pred), d(loc), d(pt), None if pw is None else d(pw), d(lt), d(lm)]
    _lib.check(lib.dgp_dlc_loss_fwd_bwd(*[None if a is None else _ptr(a) for a in args], nt, H, W, nj, 0.05, 1, _ptr(dp),
                                        _ptr(dl), _ptr(losses), _ptr(scratch), 32, None))
    torch.cuda.synchronize()
    got = losses.cpu().numpy()
    assert abs(got[0] - float(L["part_loss"])) < 1e-5 * max(1, float(L["part_loss"]))
    assert abs(got[1] - float(L["locref_loss"])) < 1e-5
    assert abs(got[2] - float(L["total_loss"])) < 1e-5 * max(1, float(L["total_loss"]))
    assert np.abs(dp.cpu().numpy() - tp.grad.numpy()).max() < 1e-6 * max(1e-3, np.abs(tp.grad.numpy()).max()) + 1e-10
    assert np.abs(dl.cpu().numpy() - tl.grad.numpy()).max() < 1e-6 * max(1e-3, np.abs(tl.grad.numpy()).max()) + 1e-10
    # empty mask: locref loss 0, gradient 0 (div_no_nan)
    z = torch.zeros_like(d(lm))
    _lib.check(lib.dgp_dlc_loss_fwd_bwd(_ptr(args[0]), _ptr(args[1]), _ptr(args[2]), None, _ptr(args[4]), _ptr(z), nt, H, W, nj,
                                        0.05, 1, _ptr(dp), _ptr(dl), _ptr(losses), _ptr(scratch), 32, None))
    torch.cuda.synchronize()
    assert float(losses[1]) == 0.0 and float(dl.abs().max()) == 0.0


def test_dlc_steps_with_changing_frame_size_match_oracle(lib_built):
    """fit_dlc inner loop: every iteration has another frame size (scale jitter / crop); two momentum steps without
    clipping vs the fp32 torch oracle, then a third size for a forward-only parity check of the re-planned net."""
    from deepgraphpose_amd import synthetic
    from deepgraphpose_amd.train import Trainer
    from oracle import dgp_train_oracle as T
    nj = 3
    wts = synthetic.make_weights(50, nj, True, seed=11, head_std=0.05)
    rng = np.random.default_rng(3)
    tr = Trainer(50, nj, 64, 64, max_frames=1)
    tr.load_weights(wts)
    P = T.make_params(wts, torch.float32)
    V = {}
    for it, (H, W) in enumerate([(70, 101), (48, 64)]):
        frames = synthetic.make_frames(1, H, W, nj, seed=it)
        sc, lmap, lmask = _dlc_targets(rng, H, W, nj)
        tr.set_input_size(H, W)
        losses = tr.forward_backward_dlc(torch.from_numpy(frames).cuda(), sc, lmap, lmask, locref_loss_weight=0.05)
        gn = tr.apply_gradients(0.005, 0.9, clip_norm=0.0)
        pred, loc = T.network(frames, P, 50, torch.float32)
        assert tuple(pred.shape) == sc.shape
        L = T.dlc_loss(pred, loc, torch.from_numpy(sc), torch.from_numpy(lmap), torch.from_numpy(lmask), None, 0.05)
        L["total_loss"].backward()
        gn_ref = T.momentum_step(P, V, 0.005, clip=0.0)
        assert abs(losses["total_loss"] - float(L["total_loss"].detach())) < 2e-4 * max(1.0, float(L["total_loss"].detach()))
        assert abs(gn - gn_ref) < 3e-3 * gn_ref
    w = tr.get_weights()
    # the second frame is 48 x 64: block3/4 run on 3 x 4 = 12 pixels, where one ReLU gate that lands on the other side of zero
    # (fp32 summation order differs between the HIP kernels and torch) moves a layer's gradient by percents; weights after two
    # steps therefore agree to 5e-4 of the tensor maximum, and the gradient norm (asserted above) to 3e-3
    for k, t in P.items():
        ref = t.detach().numpy()
        assert np.abs(w[k].reshape(ref.shape) - ref).max() <= 5e-4 * (np.abs(ref).max() + 1e-6) + 1e-6, k


def test_fit_dlc_driver(lib_built, tmp_path):
    """fit_dlc (step 0) on the synthetic project from a V1 'ImageNet' checkpoint written by tf_checkpoint: runs, logs,
    writes snapshot-step0-final--0, and fit_dgp_labeledonly can start from it."""
    import os, random
    from _project import make_project
    from deepgraphpose_amd import tf_checkpoint, weights_io
    from deepgraphpose_amd.models.fitdgp import fit_dlc, fit_dgp_labeledonly
    from deepgraphpose_amd.models.fitdgp_util import get_snapshot_path
    proj, frames, wts = make_project(tmp_path, hw=(96, 128))
    snap0, _ = get_snapshot_path("snapshot-step0-final--0", proj, shuffle=1)
    for ext in (".index", ".data-00000-of-00001"):
        os.remove(snap0 + ext)                                  # make_project seeds one; step 0 must create it
    pre = tmp_path / "pretrained"
    pre.mkdir()
    # N1 (b): the ImageNet checkpoint is a V1 tensor-slice file assembled by the INDEPENDENT code of tests/_kat_ckpt.py (snappy blocks,
    # one data block per large tensor, ordered-code keys) and holds what slim's resnet_v1_50.ckpt holds beside the backbone: the
    # classifier, mean_rgb and an int64 global_step.  fit_dlc restores the backbone only (fitdgp.py:139-142) and creates fresh heads.
    import _kat_ckpt as kat
    backbone = {k: v for k, v in wts.items() if k.startswith("resnet_v1_50")}
    backbone["resnet_v1_50/logits/weights"] = np.zeros((1, 1, 2048, 1000), dtype=np.float32)
    backbone["resnet_v1_50/logits/biases"] = np.zeros(1000, dtype=np.float32)
    backbone["resnet_v1_50/mean_rgb"] = np.array([123.68, 116.78, 103.94], dtype=np.float32)
    backbone["global_step"] = np.array(0, dtype=np.int64)
    backbone["pose/part_pred/block4/weights"] = np.full((3, 3, 3, 2048), 7.0, dtype=np.float32)     # must NOT be restored
    n_blocks = kat.v1_file(str(pre / "resnet_v1_50.ckpt"), backbone, compress=True, block_bytes=256 << 10)
    assert n_blocks >= 2
    restored = weights_io.load_weights(str(pre / "resnet_v1_50.ckpt"))
    assert "global_step" not in restored and all(np.array_equal(restored[k], v) for k, v in wts.items() if k.startswith("resnet_v1_50"))
    os.environ["DGP_PRETRAINED_DIR"] = str(pre)
    try:
        np.random.seed(0); random.seed(0)
        fit_dlc("resnet_v1_50.ckpt", proj, shuffle=1, step=0, saveiters=3, displayiters=2, maxiters=5)
    finally:
        del os.environ["DGP_PRETRAINED_DIR"]
    assert os.path.isfile(snap0 + ".index") and os.path.isfile(os.path.join(os.path.dirname(snap0), "snapshot-step0--3.index"))
    w0 = weights_io.load_weights(snap0)
    assert all(np.isfinite(v).all() for v in w0.values())
    assert np.abs(w0["resnet_v1_50/conv1/weights"] - wts["resnet_v1_50/conv1/weights"]).max() > 0
    # backbone restored from the V1 file (five momentum steps of lr <= 0.005 move a weight by far less than its scale) ...
    k3 = "resnet_v1_50/block3/unit_2/bottleneck_v1/conv2/weights"
    assert np.abs(w0[k3] - wts[k3]).max() < 0.05 * np.abs(wts[k3]).max()
    # ... heads fresh (glorot-uniform, |w| <= sqrt(3 / ((27 + 18432) / 2)) = 0.018; the file's 7.0s were not restored), both heads present
    assert np.abs(w0["pose/part_pred/block4/weights"]).max() < 0.1 and w0["pose/locref_pred/block4/weights"].shape == (3, 3, 6, 2048)
    assert sorted(k for k in w0 if "logits" in k or "mean_rgb" in k or k == "global_step") == []
    stats = open(os.path.join(os.path.dirname(snap0), "learning_stats.csv")).read().strip().splitlines()
    assert len(stats) == 2 and stats[0].startswith("iteration: 2, loss: total loss ")
    assert fit_dlc("resnet_v1_50.ckpt", proj, maxiters=5) is None            # skip-if-exists guard (fitdgp.py:113-117)
    fit_dgp_labeledonly("snapshot-step0-final--0", proj, shuffle=1, step=1, maxiters=2, displayiters=1, aug=False)


@pytest.mark.parametrize("tier", ["parity", "f16"])
def test_data_parallel_two_ranks_on_one_gpu(lib_built, tmp_path, tier):
    """N4 on hardware with W = 2: two processes on cuda:0 (control plane on gloo via DGP_DIST_BACKEND; RCCL refuses two ranks per
    device) run Trainer.step on DIFFERENT windows from the same weights.  The gradient both apply is the mean of the two
    single-process gradients, and the replicas stay bit-identical after the steps.  Round 6: the all-reduce runs GROUP BY GROUP behind the
    events the backward pass records (dgp_trainer_grad_groups / _grad_group_wait: heads + block4 first, stem last, no host
    synchronisation before the optimiser) -- the groups tile the flat buffer, the first one holds >= a quarter of the parameters; on
    the 16-bit tier the failure flag is all-reduced too (a failed prediction on one rank must skip the update on every rank)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import test_train_gpu as TT
from deepgraphpose_amd import dist as ddist
from deepgraphpose_amd.train import Trainer
from deepgraphpose_amd.loss import DGPHyper
rank, local, world = ddist.init_from_env()
import torch.distributed as dist
assert world == 2 and dist.get_backend() == "gloo"
wts = TT._train_case(7)[2]
batch, S0, _, frames, ws, ws_max = TT._train_case(20 + rank)
tr = Trainer(50, 3, 64, 96, max_frames=3, tier=sys.argv[3])
tr.load_weights(wts)
hy = DGPHyper(gm2=1, gm3=3, lr=0.005)
ft = torch.from_numpy(frames).cuda()
assert tr.grad_groups() == []                       # (no backward pass yet)
tr.step(ft, batch, hy, S0, ws, ws_max, 300.0, 25.0)
g = tr.get_grads()
groups = tr.grad_groups()
assert len(groups) >= 2 and groups[0][1] == tr.n_trainable and groups[-1][0] == 0, groups
assert all(groups[k][0] == groups[k + 1][1] for k in range(len(groups) - 1)), groups       # back to front, no gap, no overlap
assert groups[0][1] - groups[0][0] >= tr.n_trainable // 4, groups
for _ in range(2):
    tr.step(ft, batch, hy, S0, ws, ws_max, 300.0, 25.0)
if sys.argv[3] == "f16":
    assert tr.fast_passes >= 2 and tr.fast_redos == 0, (tr.fast_passes, tr.fast_redos)
w = tr.get_weights()
np.savez(sys.argv[2] + "_g%d.npz" % rank, **g)
np.savez(sys.argv[2] + "_w%d.npz" % rank, **w)
dist.barrier()
dist.destroy_process_group()
'''
    procs = []
    for rank in (0, 1):
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), RANK=str(rank), WORLD_SIZE="2",
                   LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", DGP_DIST_BACKEND="gloo")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-c", code, os.path.join(root, "tests"), str(tmp_path / "dp"), tier], env=env, cwd=root,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for pr in procs:
        try:
            outs.append(pr.communicate(timeout=900))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for pr, (so, se) in zip(procs, outs):
        assert pr.returncode == 0, se[-2000:]
    g = [np.load(str(tmp_path / "dp") + "_g%d.npz" % r) for r in (0, 1)]
    w = [np.load(str(tmp_path / "dp") + "_w%d.npz" % r) for r in (0, 1)]
    for k in g[0].files:
        assert np.array_equal(g[0][k], g[1][k]), k            # the all-reduce leaves the same bits on every rank
    for k in w[0].files:
        assert np.array_equal(w[0][k], w[1][k]), k            # ... so the replicas never diverge
    # the averaged gradient = mean of what each window gives alone
    from deepgraphpose_amd.train import Trainer
    from deepgraphpose_amd.loss import DGPHyper
    wts = _train_case(7)[2]
    single = []
    for r in (0, 1):
        batch, S0, _, frames, ws, ws_max = _train_case(20 + r)
        tr = Trainer(50, 3, 64, 96, max_frames=3)
        tr.load_weights(wts)
        tr.forward_backward(torch.from_numpy(frames).cuda(), batch, DGPHyper(gm2=1, gm3=3, lr=0.005), S0, ws, ws_max, 300.0, 25.0)
        single.append(tr.get_grads())
    for k in g[0].files:
        ref = 0.5 * (single[0][k].astype(np.float64) + single[1][k].astype(np.float64))
        tol = 2e-5 * (np.abs(ref).max() + 1e-12) + 1e-9
        assert np.abs(g[0][k] - ref).max() <= tol, (k, float(np.abs(g[0][k] - ref).max()), float(np.abs(ref).max()))


def test_fit_dgp_driver_two_ranks_on_one_gpu(lib_built, tmp_path):
    """The fit driver under the launcher's environment with W = 2 (both ranks on cuda:0, gloo control plane): the ranks join the
    group before touching the GPU, share rank 0's schedule seed, take entries it W + r of the schedule, average gradients every
    step, and rank 0 alone writes the step snapshots."""
    import os, subprocess, sys
    from _project import make_project
    from deepgraphpose_amd.models.fitdgp_util import get_snapshot_path
    from deepgraphpose_amd import weights_io
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proj, frames, wts = make_project(tmp_path)
    code = r'''
import os, sys, numpy as np, random
np.random.seed(int(os.environ["RANK"])); random.seed(int(os.environ["RANK"]))      # different local seeds: rank 0's must win
from deepgraphpose_amd.models.fitdgp import fit_dgp
fit_dgp("snapshot-step0-final--0", sys.argv[1], batch_size=4, shuffle=1, step=2, maxiters=6, displayiters=1, gm2=1, gm3=3,
        aug=False, n_max_frames=30, ns=3)
import torch.distributed as dist
assert dist.is_initialized() and dist.get_world_size() == 2
dist.barrier()
dist.destroy_process_group()
print("rank", os.environ["RANK"], "done")
'''
    procs = []
    for rank in (0, 1):
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), RANK=str(rank), WORLD_SIZE="2",
                   LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29643", DGP_DIST_BACKEND="gloo")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-c", code, proj], env=env, cwd=root, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for pr in procs:
        try:
            outs.append(pr.communicate(timeout=900))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for pr, (so, se) in zip(procs, outs):
        assert pr.returncode == 0, (so[-500:], se[-2000:])
    snap2, _ = get_snapshot_path("snapshot-step2-final--0", proj, shuffle=1)
    assert os.path.isfile(snap2 + ".index")
    w2 = weights_io.load_weights(snap2)
    assert all(np.isfinite(v).all() for v in w2.values())
    assert np.abs(w2["pose/part_pred/block4/weights"] - wts["pose/part_pred/block4/weights"]).max() > 0
    # 6 schedule entries over 2 ranks = 3 optimiser steps: the iteration snapshot rank 0 wrote is the third
    assert os.path.isfile(snap2.replace("-final--0", "--3") + ".index") or os.path.isfile(snap2.replace("-final--0", "--2") + ".index")


def test_estimate_from_a_saver_bundle_with_slots_and_global_step(lib_built, tmp_path):
    """N1 (c): a snapshot as tf.train.Saver() writes it after the optimiser exists -- every model variable, `<var>/Momentum` slots and
    the int64 `global_step` -- assembled by the independent code of tests/_kat_ckpt.py (snappy index blocks).  The extra variables
    are skipped, not rejected, and the engine loaded from that prefix matches the oracle on the same model variables."""
    import _kat_ckpt as kat
    from deepgraphpose_amd import engine, synthetic, weights_io
    from deepgraphpose_amd.models.eval import setup_dgp_eval_graph
    from oracle import dgp_oracle as O
    nj, B, h, w = 3, 4, 96, 128
    wts = synthetic.make_weights(50, nj, True, seed=21, head_std=0.05)
    rng = np.random.default_rng(0)
    extra = {k + "/Momentum": rng.standard_normal(v.shape).astype(np.float32) for k, v in wts.items()
             if "moving_" not in k and v.size < 300000}
    extra["global_step"] = np.array(200000, dtype=np.int64)
    extra["beta1_power"] = np.float32(0.9).reshape(())                  # a float scalar no layer owns (Adam leaves such behind)
    prefix = str(tmp_path / "snapshot-step0-final--0")
    assert kat.bundle(prefix, {**wts, **extra}, compress=True, block_bytes=4096) >= 3
    loaded = weights_io.load_weights(prefix)
    assert not any(k.endswith("/Momentum") or k == "global_step" for k in loaded)
    frames = synthetic.make_frames(B, h, w, nj, seed=5)
    from types import SimpleNamespace
    cfg = dict(net_type="resnet_50", num_joints=nj)
    dlc_cfg = SimpleNamespace(**cfg, get=lambda k, d=None: cfg.get(k, d))
    sess, mu_n, _, scmap, _, inputs = setup_dgp_eval_graph(dlc_cfg, prefix)          # Saver.restore(sess, prefix), eval.py:194-211
    mu, sc = sess.run([mu_n, scmap], feed_dict={inputs: frames})
    ref = O.infer(frames, wts, 50, 8.0, 1.0, 1)
    assert np.abs(np.asarray(mu) - ref["mu"]).max() * 8.0 < 1e-3
    assert np.abs(np.asarray(sc) - ref["scmap"]).max() < 1e-4 * np.abs(ref["scmap"]).max()
    net = engine.DGPNet(50, nj, h, w, max_batch=B, device=0)
    net.load_weights(loaded)
    _, _, idx = net.infer(torch.from_numpy(frames).cuda(), 1.0, 1)
    assert np.array_equal(idx.cpu().numpy(), ref["idx"])


@pytest.mark.parametrize("nj", [1, 2])
def test_trainer_sync_then_net_load_weights_share_one_head_panel_width(lib_built, nj):
    """The heads' pointwise panel buffers (l.d_w_pw / l.d_wh3_pw) are shared by dgp_trainer_sync_weights and dgp_net_load_weights and are
    allocated by whichever runs first: both must size them with the same rule (head_pw_coutp: one or two joints pad to 64 columns).
    Before the fix a trainer sync (32 columns) followed by net.load_weights (64) wrote 512 KB into a 256-KB buffer."""
    from deepgraphpose_amd import synthetic
    from deepgraphpose_amd.train import Trainer
    from oracle import dgp_oracle as O
    h, w = 64, 96
    wts = synthetic.make_weights(50, nj, True, seed=31, head_std=0.05)
    frames = synthetic.make_frames(2, h, w, nj, seed=1)
    tr = Trainer(50, nj, h, w, max_frames=2)
    tr.load_weights(wts)                          # uploads + dgp_trainer_sync_weights: allocates the shared panels
    tr.sync()
    tr.net.load_weights(wts)                      # the public path the advisor named: dgp_net_load_weights on the trainer's net
    ft = torch.from_numpy(frames).cuda()
    mu, conf, idx = tr.net.infer(ft, 1.0, 1)
    torch.cuda.synchronize()
    ref = O.infer(frames, wts, 50, 8.0, 1.0, 1)
    assert np.abs(mu.cpu().numpy() - ref["mu"]).max() * 8.0 < 1e-3
    assert np.array_equal(idx.cpu().numpy(), ref["idx"])
    tr.sync()                                     # and back: the trainer's pack kernels write the same buffers again
    sc = tr._forward(ft)[1]
    torch.cuda.synchronize()
    assert np.abs(sc.cpu().numpy() - ref["scmap"]).max() < 1e-4 * np.abs(ref["scmap"]).max()


def _grad_agreement(g, P):
    """global and per-tensor agreement of a gradient dict with the oracle's: cosine, relative L2 error"""
    dots, n1, n2, per = 0.0, 0.0, 0.0, {}
    for k, t in P.items():
        if not t.requires_grad:
            continue
        ref = t.grad.numpy().astype(np.float64).ravel()
        got = g[k].astype(np.float64).ravel()
        dots += float(ref @ got); n1 += float(ref @ ref); n2 += float(got @ got)
        per[k] = (float(ref @ got) / (np.linalg.norm(ref) * np.linalg.norm(got) + 1e-300), float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-300)))
    cos = dots / np.sqrt(n1 * n2)
    rel = np.sqrt(max(n1 + n2 - 2 * dots, 0.0) / n1)
    return cos, rel, per


@pytest.mark.parametrize("seed,hw,nt,nj", [(3, (64, 96), 3, 3), (5, (128, 160), 4, 3), (7, (64, 96), 3, 2), (9, (96, 64), 2, 1),
                                           (11, (102, 90), 2, 3)])      # (51 x 45 conv1 map: the pool's SAME padding starts at 1, ragged stem tiles)
def test_trainer_tier_f16_gradients_against_the_fp64_oracle(lib_built, seed, hw, nt, nj):
    """BASELINE configs[3] names bf16: the 16-bit tier of the training step (Trainer(tier="f16"); dgp_trainer_set_tier).  From the second pass
    of a shape on, blocks 2-4 keep activations and gradient tensors as 2-byte H1 cells with predicted scales, their convs run one MFMA per
    product and their weight gradients read both operands in place (wgrad_dma_h1).  (nj = 1, 2: the heads' pointwise panels are padded to
    the 64-column cell tile -- head_pw_coutp -- and both heads' backward shares one 64-channel H1 tensor.)  Against the float64 autograd oracle: loss within
    2e-3 relative, global gradient cosine >= 0.999 and relative L2 error <= 5 %, every tensor's cosine >= 0.98 -- a REPORTED tier, the
    parity tier's tolerances (2e-5 / 3e-3) are for tier 0.  The first pass of a shape runs on the parity path and meets THOSE."""
    import ctypes
    from deepgraphpose_amd.train import Trainer
    from deepgraphpose_amd.loss import DGPHyper
    batch, S0, wts, frames, ws, ws_max = _train_case(seed, hw=hw, nt=nt, nj=nj)
    hy = DGPHyper(gm2=1, gm3=3)
    n_tot, n_vis = 300.0, 25.0
    P, L = _oracle_grads(wts, frames, batch, S0, ws, ws_max, hy, n_tot, n_vis, dtype=torch.float64)
    ref_loss = float(L["total_loss"].detach())
    tr = Trainer(50, nj, hw[0], hw[1], max_frames=nt, tier="f16")
    tr.load_weights(wts)
    ft = torch.from_numpy(frames).cuda()
    was, failed = ctypes.c_int32(), ctypes.c_int32()
    l0 = tr.forward_backward(ft, batch, hy, S0, ws, ws_max, n_tot, n_vis)            # first pass of the shape: parity path
    tr.lib.dgp_trainer_fast_status(tr._t, was, failed)
    assert was.value == 0 and abs(l0["total_loss"] - ref_loss) < 1e-4 * max(1, abs(ref_loss))
    cos0, rel0, _ = _grad_agreement(tr.get_grads(), P)
    assert cos0 > 0.99999 and rel0 < 3e-3
    l1 = tr.forward_backward(ft, batch, hy, S0, ws, ws_max, n_tot, n_vis)            # 16-bit pass
    tr.lib.dgp_trainer_fast_status(tr._t, was, failed)
    assert was.value == 1 and failed.value == 0 and tr.fast_redos == 0 and tr.fast_passes == 1
    cos1, rel1, per = _grad_agreement(tr.get_grads(), P)
    worst = sorted(per.items(), key=lambda kv: kv[1][0])[:3]
    print("tier f16 %s nt %d: loss %.6f (fp64 %.6f) | gradient cosine %.6f rel L2 %.4f | worst tensors %s" % (hw, nt, l1["total_loss"], ref_loss, cos1, rel1, worst))
    assert abs(l1["total_loss"] - ref_loss) < 2e-3 * max(1, abs(ref_loss))
    assert cos1 >= 0.999 and rel1 <= 0.05
    assert min(v[0] for v in per.values()) >= 0.98, worst
    assert np.all([np.isfinite(v).all() for v in tr.get_grads().values()])
    l2 = tr.forward_backward(ft, batch, hy, S0, ws, ws_max, n_tot, n_vis)            # deterministic up to the float atomics of dW
    assert abs(l2["total_loss"] - l1["total_loss"]) < 1e-6 * max(1, abs(l1["total_loss"]))


def test_trainer_tier_f16_recovers_from_a_failed_prediction(lib_built):
    """The 16-bit tier's tensors have no fp32 twins: when the ranges jump by more than the predicted scales cover (new weights 64 x larger
    in the stem, smuggled in behind the wrapper's back) the step raises the flag and is repeated on the parity path before anything is
    returned; the step after that is a 16-bit pass again."""
    import ctypes
    from deepgraphpose_amd.train import Trainer
    from deepgraphpose_amd.loss import DGPHyper
    batch, S0, wts, frames, ws, ws_max = _train_case(3)
    hy = DGPHyper(gm2=1, gm3=3)
    tr = Trainer(50, 3, 64, 96, max_frames=3, tier="f16")
    tr.load_weights(wts)
    ft = torch.from_numpy(frames).cuda()
    was, failed = ctypes.c_int32(), ctypes.c_int32()
    tr.forward_backward(ft, batch, hy, S0, ws, ws_max, 300.0, 25.0)
    tr.forward_backward(ft, batch, hy, S0, ws, ws_max, 300.0, 25.0)
    assert tr.fast_passes == 1 and tr.fast_redos == 0
    w2 = dict(wts)
    w2["resnet_v1_50/conv1/BatchNorm/gamma"] = wts["resnet_v1_50/conv1/BatchNorm/gamma"] * 64.0
    w2["resnet_v1_50/conv1/BatchNorm/beta"] = wts["resnet_v1_50/conv1/BatchNorm/beta"] * 64.0
    key = tr._fast_key
    tr.load_weights(w2)
    tr._fast_key = key                                           # pretend nothing happened: the next pass is requested as a 16-bit one
    losses = tr.forward_backward(ft, batch, hy, S0, ws, ws_max, 300.0, 25.0)
    assert tr.fast_redos == 1
    tr.lib.dgp_trainer_fast_status(tr._t, was, failed)
    assert was.value == 0                                        # what was returned came from the parity path ...
    P, L = _oracle_grads(w2, frames, batch, S0, ws, ws_max, hy, 300.0, 25.0, dtype=torch.float64)
    cos, rel, _ = _grad_agreement(tr.get_grads(), P)
    assert cos > 0.99999 and rel < 3e-3                          # ... and meets ITS tolerances
    assert abs(losses["total_loss"] - float(L["total_loss"].detach())) < 1e-4 * max(1, abs(float(L["total_loss"].detach())))
    tr.forward_backward(ft, batch, hy, S0, ws, ws_max, 300.0, 25.0)
    tr.lib.dgp_trainer_fast_status(tr._t, was, failed)
    assert was.value == 1 and failed.value == 0 and tr.fast_redos == 1
    cos, rel, _ = _grad_agreement(tr.get_grads(), P)
    assert cos >= 0.999 and rel <= 0.05


def test_fifty_steps_of_both_trainer_tiers_stay_in_one_band(lib_built):
    """50 optimiser steps (clip-by-global-norm + momentum, lr 0.005) from the same weights on the same batch, parity tier and 16-bit tier:
    both losses fall, and the 16-bit trajectory stays within 3 % (+ 2e-3) of the parity trajectory at every step -- the stated band of
    the reported tier; no step of the 16-bit run needed a repeat."""
    from deepgraphpose_amd.train import Trainer
    from deepgraphpose_amd.loss import DGPHyper
    batch, S0, wts, frames, ws, ws_max = _train_case(11, hw=(96, 128), nt=4)
    hy = DGPHyper(gm2=1, gm3=3)
    ft = torch.from_numpy(frames).cuda()
    curves = {}
    for tier in ("parity", "f16"):
        tr = Trainer(50, 3, 96, 128, max_frames=4, tier=tier)
        tr.load_weights(wts)
        curves[tier] = np.array([tr.step(ft, batch, hy, S0, ws, ws_max, 300.0, 25.0)["total_loss"] for _ in range(50)])
        if tier == "f16":
            assert tr.fast_passes == 49 and tr.fast_redos == 0
    a, b = curves["parity"], curves["f16"]
    print("loss step 0 / 10 / 25 / 49: parity %.4f %.4f %.4f %.4f | f16 %.4f %.4f %.4f %.4f | max relative gap %.4f" % (
        a[0], a[10], a[25], a[49], b[0], b[10], b[25], b[49], float(np.max(np.abs(a - b) / np.abs(a)))))
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert a[49] < 0.8 * a[0] and b[49] < 0.8 * b[0]
    assert (np.abs(a - b) <= 0.03 * np.abs(a) + 2e-3).all()


_STALE_PANEL_CHILD = r"""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import test_train_gpu as TT
from deepgraphpose_amd import engine
from deepgraphpose_amd.train import Trainer
from deepgraphpose_amd.loss import DGPHyper
batch, S0, wts, frames, ws, ws_max = TT._train_case(13, hw=(64, 96), nt=3, nj=3)
hy = DGPHyper(gm2=1, gm3=3, lr=0.02)
tr = Trainer(50, 3, 64, 96, max_frames=3, tier="f16")
tr.load_weights(wts)
ft = torch.from_numpy(frames).cuda()
for _ in range(3):                                  # optimiser steps: the weights move, every sync after the first is a lazy one
    tr.step(ft, batch, hy, S0, ws, ws_max, 300.0, 25.0)
assert tr.fast_passes == 2 and tr.fast_redos == 0
w_now = tr.get_weights()
assert max(float(np.abs(w_now[k] - wts[k]).max()) for k in wts if "pose/" in k and "weights" in k) > 1e-4      # the heads DID move
was, failed = ctypes.c_int32(), ctypes.c_int32()
l = tr.forward_backward(ft, batch, hy, S0, ws, ws_max, 300.0, 25.0)           # a 16-bit pass on the CURRENT weights
tr.lib.dgp_trainer_fast_status(tr._t, was, failed)
assert was.value == 1 and failed.value == 0
P, L = TT._oracle_grads(w_now, frames, batch, S0, ws, ws_max, hy, 300.0, 25.0, dtype=torch.float64)
cos, rel, per = TT._grad_agreement(tr.get_grads(), P)
worst = sorted(per.items(), key=lambda kv: kv[1][0])[:3]
print("heads merged=%s: cosine %.6f rel %.4f worst %s" % (sys.argv[2], cos, rel, worst))
assert abs(l["total_loss"] - float(L["total_loss"].detach())) < 2e-3 * max(1, abs(float(L["total_loss"].detach())))
assert cos >= 0.999 and rel <= 0.05 and min(v[0] for v in per.values()) >= 0.98, (cos, rel, worst)
# a forward on the trainer-owned net between steps reads the heads' 2x2-conv panels: they must be the CURRENT weights'
mu_t, conf_t, idx_t = tr.net.infer(ft, 1.0, 1)
fresh = engine.DGPNet(50, 3, 64, 96, max_batch=3, with_locref=True)
fresh.load_weights(w_now)
mu_f, conf_f, idx_f = fresh.infer(ft, 1.0, 1)
d = float((mu_t - mu_f).abs().max()) * 8.0
print("trainer-owned net vs a fresh net with the same weights: %.3g px" % d)
assert d < 1e-3, d
"""


@pytest.mark.parametrize("heads_h1", ["0", "1"])
def test_trainer_tier_f16_never_reads_stale_head_panels(lib_built, heads_h1):
    """Round-5 advisor finding: in the 16-bit tier every weight sync after the first skips the parity-only panels (the heads' 2x2-conv and
    per-head data-gradient panels, the H2 cells).  With the merged-heads H1 panel switched off (DGP_TRAIN_HEADS_H1=0, read once per
    process -> child process) a 16-bit backward takes the per-head branch and read the data-gradient panels of the LAST PLAIN pass; and
    a forward on the trainer-owned net read the stale 2x2-conv panels.  Now both refresh first: gradients after three optimiser steps
    against the fp64 oracle ON THE CURRENT WEIGHTS, and Trainer.net.infer against a fresh net with those weights."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _STALE_PANEL_CHILD, root, heads_h1], env=dict(os.environ, DGP_TRAIN_HEADS_H1=heads_h1, PYTHONPATH=root),
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
