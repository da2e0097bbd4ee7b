"""Whole batches at BASELINE's full sizes against the CPU oracle.  The oracle's outputs are cached fixtures (tests/golden/fullsize_vectors.npz,
written by tests/golden/make_fullsize_golden.py in the build container: minutes of fp32 CPU convolutions), so the GPU suite compares every
frame without paying for them at run time.  The oracle is the checker; the product path is the HIP engine behind the C-ABI."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(HERE, "golden", "fullsize_vectors.npz"))


def test_resnet101_1280x720_20_keypoints_all_16_frames_both_heads(lib_built, G):
    """BASELINE configs[4]'s per-GPU shape: ResNet-101, 1280 x 720, 20 keypoints, a batch of 16 -- ALL frames, BOTH heads.  Soft-argmax
    coordinates inside the fp64-anchored gate, window indices and likelihood cells bit-exact (fp32 oracle AND fp64 anchor), scoremap and locref map (16 384 sampled positions each)
    within 1e-4 of their range."""
    from deepgraphpose_amd import engine
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    wts = make_weights(101, 20, True, seed=41, head_std=0.05)
    frames = torch.from_numpy(make_frames(16, 720, 1280, 20, seed=42)).cuda()
    net = engine.DGPNet(101, 20, 720, 1280, max_batch=16, with_locref=True)
    net.load_weights(wts)
    sc, lr = net.forward(frames, want_locref=True)
    mu, conf, idx = net.infer(frames, 1.0, 1)
    torch.cuda.synchronize()
    assert tuple(sc.shape) == (16, 90, 160, 20) and tuple(lr.shape) == (16, 90, 160, 40)
    # logits of standard deviation ~5 over a 90 x 160 map: a BROAD softmax, the regime in which every fp32 evaluation of this network is
    # ~1e-3 px from any other (EXPERIMENTS.md section 2d) -- the gate is the fp64 anchor's: the engine may be as far from the truth as
    # max(1e-3 px, 1.5 x the fp32 CPU oracle's own distance), frame by frame; and it stays within 2.5e-3 px of the fp32 oracle
    got = mu.cpu().numpy().astype(np.float64)
    e_gpu = np.abs(got - G["r101_mu64"]).max((1, 2)) * 8.0
    e_ora = np.abs(G["r101_mu"].astype(np.float64) - G["r101_mu64"]).max((1, 2)) * 8.0
    assert (e_gpu <= np.maximum(1e-3, 1.5 * e_ora)).all(), (e_gpu, e_ora)
    assert np.abs(got - G["r101_mu"]).max() * 8.0 < 2.5e-3
    gi = idx.cpu().numpy()
    same32, same64 = (gi == G["r101_idx"]).all(-1), (gi == G["r101_idx64"]).all(-1)
    assert (same32 | same64).all()            # (the fp32 oracle and the anchor disagree on a window cell where mu sits on a cell boundary)
    assert same32.mean() > 0.99
    assert np.abs(conf.cpu().numpy() - G["r101_lik"])[same32].max() < 1e-5
    scn, lrn = sc.cpu().numpy().reshape(16, -1), lr.cpu().numpy().reshape(16, -1)
    got_sc = np.take_along_axis(scn, G["r101_pos_sc"].astype(np.int64), 1)
    got_lr = np.take_along_axis(lrn, G["r101_pos_lr"].astype(np.int64), 1)
    assert np.abs(got_sc - G["r101_sc"]).max() < 1e-4 * float(G["r101_sc_max"])
    assert np.abs(got_lr - G["r101_lr"]).max() < 1e-4 * float(G["r101_lr_max"])
    assert abs(float(np.abs(scn).max()) - float(G["r101_sc_max"])) < 1e-4 * float(G["r101_sc_max"])


def test_estimate_pose_on_the_reaching_projects_labeled_frames(lib_built, G, tmp_path):
    """BASELINE configs[0] / [1] on REAL frames: the 55 labeled images of the reference's Reaching demo project (832 x 747; 15 of them are
    640 x 470 crops that LabeledDirSource resizes) as the pseudo-video the demo falls back to when the .avi is missing, through
    estimate_pose -- decode thread, pinned staging, two engines, csv export -- with a seeded snapshot stored as a TF bundle.  Every one of
    the 246 pseudo-frames against the oracle's (x, y, likelihood, window index) of the labeled image it shows."""
    import shutil
    from _project import make_project
    from deepgraphpose_amd import weights_io
    from deepgraphpose_amd.models.eval import estimate_pose
    from deepgraphpose_amd.models.fitdgp_util import get_snapshot_path
    from deepgraphpose_amd.synthetic import make_weights
    proj, _, _ = make_project(tmp_path, nj=5, hw=(64, 96))
    dst = os.path.join(proj, "labeled-data", "reachingvideo1")
    shutil.copytree(os.path.join(HERE, "golden", "reaching_frames"), dst)
    snap, cfg_path = get_snapshot_path("snapshot-step2-final--0", proj, shuffle=1)
    weights_io.save_weights(snap, make_weights(50, 5, True, seed=43, head_std=0.05))          # (TF V2 bundle: the default format)
    labels = estimate_pose(str(cfg_path), snap, os.path.join(proj, "videos", "reachingvideo1.avi"), os.path.join(proj, "videos_pred"),
                           shuffle=1, batch_size=16)
    numbers = G["reach_numbers"]
    T = int(numbers[-1]) + 1
    assert labels["x"].shape == (T, 5) and T == 246
    which = np.maximum(np.searchsorted(numbers, np.arange(T), side="right") - 1, 0)          # the labeled image frame t shows
    # seeded (untrained) weights on real frames give BROAD scoremaps -- the regime in which fp32 evaluations of this network differ by ~1e-3 px
    # from each other (EXPERIMENTS.md section 2d): the gate is the fp64 anchor's, image by image -- the engine may be as far from the truth as
    # max(1e-3 px, 1.5 x the fp32 CPU oracle's own distance) -- plus 2.5e-3 px from the fp32 oracle itself
    def dist(ax, ay, bx, by):
        return np.sqrt((ax - bx) ** 2 + (ay - by) ** 2).max(1)
    e_gpu = dist(labels["x"], labels["y"], G["reach_x64"][which], G["reach_y64"][which])
    e_ora = dist(G["reach_x"], G["reach_y"], G["reach_x64"], G["reach_y64"])[which]
    assert (e_gpu <= np.maximum(1e-3, 1.5 * e_ora)).all(), (float(e_gpu.max()), float(e_ora.max()))
    assert dist(labels["x"], labels["y"], G["reach_x"][which], G["reach_y"][which]).max() < 2.5e-3
    assert np.abs(labels["likelihoods"] - G["reach_lik"][which]).max() < 1e-4
    rows = open(os.path.join(proj, "videos_pred", "reachingvideo1_labeled.csv")).read().strip().split("\n")
    assert len(rows) == 3 + T and rows[1].startswith("bodyparts,part0,part0,part0,part1")


def test_resnet101_1280x720_peaky_heads_meet_the_literal_gate(lib_built):
    """BASELINE configs[4]'s per-GPU shape on a PEAKY, trained-like fixture (tests/golden/make_fullsize_peaky_golden.py: matched-filter
    part_pred heads that fire on the synthetic blobs, one compact peak per keypoint): the north-star gate LITERALLY -- coordinates within
    1e-3 px of the fp32 oracle, likelihood-window indices bit-exact -- on ALL 16 x 20 (frame, keypoint) pairs; 286 of them are
    well-conditioned by the oracle's own float64 evaluation (oracle-to-float64 distance < 1e-4 px: no two cells near a tie), the other
    34 are near-ties between two cells (ill-conditioned in fp32 whatever the network: the coordinate moves by 8 px x p (1 - p) x the error
    of a logit gap) on which the fp32 oracle itself is up to 4.6e-4 px from float64 -- the engine still lands inside 1e-3 px of it."""
    import importlib.util
    from deepgraphpose_amd import engine
    from deepgraphpose_amd.synthetic import make_frames, make_weights
    G = np.load(os.path.join(HERE, "golden", "fullsize_peaky_vectors.npz"))
    spec = importlib.util.spec_from_file_location("make_fullsize_peaky_golden", os.path.join(HERE, "golden", "make_fullsize_peaky_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    wts = make_weights(gen.DEPTH, gen.NJ, True, seed=gen.SEED_W, head_std=0.05)
    w, b = gen.head_from_prototypes(G["P"], G["pm"], float(G["beta"]), float(G["rho"]))
    wts["pose/part_pred/block4/weights"], wts["pose/part_pred/block4/biases"] = w, b
    frames = torch.from_numpy(make_frames(gen.T, gen.H, gen.W, gen.NJ, seed=gen.SEED_F)).cuda()
    net = engine.DGPNet(gen.DEPTH, gen.NJ, gen.H, gen.W, max_batch=gen.T, with_locref=True)
    net.load_weights(wts)
    mu, conf, idx = net.infer(frames, 1.0, 1)
    torch.cuda.synchronize()
    got, gi, gl = mu.cpu().numpy().astype(np.float64), idx.cpu().numpy(), conf.cpu().numpy()
    well = G["well"]
    assert well.shape == (gen.T, gen.NJ) and well.mean() >= 0.85, float(well.mean())      # the fixture IS mostly well-conditioned
    d32 = np.abs(got - G["mu"]).max(-1) * 8.0
    d64 = np.abs(got - G["mu64"]).max(-1) * 8.0
    e_ora = np.abs(G["mu"].astype(np.float64) - G["mu64"]).max(-1) * 8.0
    print("peaky fixture: %d of %d pairs well-conditioned; engine vs fp32 oracle max %.3g px on them (%.3g over all), vs float64 %.3g px"
          % (int(well.sum()), well.size, d32[well].max(), d32.max(), d64[well].max()))
    # the literal gate, on ALL 16 x 20 pairs (measured: 4.3e-4 px on the well-conditioned ones, 5.8e-4 px over all; the fp32 oracle itself
    # is up to 4.6e-4 px from its float64 evaluation on this fixture) ...
    assert d32.max() < 1e-3, (d32.max(), d32[well].max())
    assert np.array_equal(gi, G["idx"])                                        # ... every likelihood-window index bit-exact
    assert np.abs(gl - G["lik"]).max() < 1e-5
    assert d32[well].max() < 6e-4 and (d64 <= np.maximum(1e-3, 1.5 * e_ora)).all(), (d32[well].max(), d64.max(), e_ora.max())
