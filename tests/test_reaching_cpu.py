"""SURVEY.md 8(c) fixture (5) and BASELINE configs[0] plumbing on the reference's own demo project (Reaching-Mackenzie-2018-08-30).

tests/golden/reaching_vectors.npz was produced by RUNNING the reference loader and the reference's own lines on the shipped .mat
(tests/golden/make_reaching_golden.py); the .mat itself is kept beside it as an input data file.  The last test opens the real
project tree and only runs where /root/reference exists (the build container).
"""
import os
import shutil

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
MAT = os.path.join(HERE, "golden", "Reaching_Mackenzie95shuffle1.mat")
REF_PROJ = "/root/reference/data/Reaching-Mackenzie-2018-08-30"
BODYPARTS = ["Hand", "Finger1", "Tongue", "Joystick1", "Joystick2"]
SKELETON = [["Hand", "Finger1"], ["Joystick1", "Joystick2"]]


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(HERE, "golden", "reaching_vectors.npz"), allow_pickle=False)


def test_skeleton_matrix_matches_reference_lines(G):
    from deepgraphpose_amd.config import skeleton_matrix
    S0 = skeleton_matrix({"bodyparts": BODYPARTS, "skeleton": SKELETON})
    assert np.array_equal(S0, G["S0"])
    assert skeleton_matrix({"bodyparts": BODYPARTS, "skeleton": None}).shape == (0, 5)


def test_train_mat_labels_match_reference_loader(G):
    """load_train_mat_labels == PoseDataset.load_dataset + the targets_2d loop of Dataset._compute_targets (dataset.py:643-652)."""
    from deepgraphpose_amd.dataset import load_train_mat_labels
    targets, frames = load_train_mat_labels(MAT, "reachingvideo1", 5, 8.0)
    order = np.argsort(G["frame_idxs"])
    assert np.array_equal(frames, G["frame_idxs"][order])
    np.testing.assert_array_equal(targets, G["targets_2d"][order])          # same arithmetic -> same bits (NaN == NaN)
    assert targets.shape == (52, 5, 2) and np.isnan(targets).any()
    t0, f0 = load_train_mat_labels(MAT, "no_such_video", 5, 8.0)
    assert t0.shape == (0, 5, 2) and f0.size == 0


def test_limb_statistics_match_reference_lines(G):
    """B1: ws, ws_max of fitdgp.py:875-892 on the real labels -- product (_limb_statistics) and oracle (limb_statistics)."""
    from deepgraphpose_amd.models.fitdgp import _limb_statistics
    from oracle.dgp_train_oracle import limb_statistics
    t = G["targets_2d"]
    for S0, ws_ref, wm_ref in ((G["S0"], G["ws"], G["ws_max"]), (G["dense_S0"], G["dense_ws"], G["dense_ws_max"])):
        ws, ws_max = _limb_statistics([t], S0, 8.0, 1000, 1.2)
        np.testing.assert_allclose(ws, ws_ref, rtol=1e-12)
        np.testing.assert_allclose(ws_max, wm_ref, rtol=1e-12)
        ws_o, wm_o = limb_statistics(t, S0, 8.0, 1000, 1.2)
        np.testing.assert_allclose(ws_o, ws_ref, rtol=1e-12)
        np.testing.assert_allclose(wm_o, wm_ref, rtol=1e-12)
    # several datasets are stacked before the statistics (order does not matter), empty ones are skipped
    ws2, wm2 = _limb_statistics([t[:20], np.empty((0, 5, 2)), t[20:]], G["S0"], 8.0, 1000, 1.2)
    np.testing.assert_allclose(ws2, G["ws"], rtol=1e-12)
    np.testing.assert_allclose(wm2, G["ws_max"], rtol=1e-12)


def test_dlc_pose_dataset_loads_the_mat_like_the_reference(G, tmp_path):
    """dlc_dataset.PoseDataset.load_dataset vs the reference's PoseDataset.load_dataset (pose_defaultdataset.py:39-76)."""
    from deepgraphpose_amd.config import AttrDict
    from deepgraphpose_amd.dlc_dataset import PoseDataset
    cfg = AttrDict(project_path=os.path.dirname(MAT), dataset=os.path.basename(MAT), num_joints=5, locref_stdev=7.2801, stride=8.0,
                   global_scale=0.8, mirror=False, shuffle=False, crop=False, pos_dist_thresh=17, location_refinement=True,
                   scale_jitter_lo=0.5, scale_jitter_up=1.25, deterministic=True, batch_size=1, cropratio=0.4, minsize=100,
                   leftwidth=400, rightwidth=400, topheight=400, bottomheight=400, weigh_only_present_joints=False)
    ds = PoseDataset(cfg)
    assert ds.num_images == int(G["n_items"]) == 52
    assert [str(d.im_path) for d in ds.data] == [str(p) for p in G["im_paths"]]
    assert np.array_equal(np.array([np.asarray(d.im_size).ravel() for d in ds.data]), G["im_sizes"])
    flat = np.concatenate([np.asarray(d.joints[0], dtype=np.float64) for d in ds.data])
    assert np.array_equal(np.array([d.joints[0].shape[0] for d in ds.data]), G["joints_len"])
    np.testing.assert_array_equal(flat, G["joints_flat"])


@pytest.mark.skipif(not os.path.isdir(REF_PROJ), reason="the reference tree only exists in the build container")
def test_real_reaching_project_plumbing(G, tmp_path, monkeypatch):
    """BASELINE configs[0] without a GPU: the demo script's config rewriting on a copy of the real project, both pose_cfg.yaml,
    the 55 labeled PNGs (57 directory entries) as a pseudo-video, the batch bookkeeping and the loss pre-computation (5 bodyparts / 2 limbs)."""
    import sys
    import yaml
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "demo"))
    import run_dgp_demo as demo
    from deepgraphpose_amd import config as dcfg
    from deepgraphpose_amd.dataset import MultiDataset
    from deepgraphpose_amd.frames import LabeledDirSource, open_frame_source
    from deepgraphpose_amd.models.fitdgp import dgp_loss, _video_sets
    from deepgraphpose_amd.models.fitdgp_util import gen_batch, get_snapshot_path
    from deepgraphpose_amd.models.session import PLACEHOLDER_KEYS

    work = tmp_path / "work"
    dst = work / demo.DEMO_PROJECT
    shutil.copytree(REF_PROJ, dst)
    for root, dirs, files in os.walk(dst):                    # the reference tree is read-only
        for n in dirs + files:
            os.chmod(os.path.join(root, n), 0o755)
    monkeypatch.chdir(work)
    before = {p: open(p).read() for p in (str(dst / "config.yaml"), demo.get_model_cfg_path(str(work), demo.DEMO_PROJECT, "train"),
                                          demo.get_model_cfg_path(str(work), demo.DEMO_PROJECT, "test"))}
    with pytest.raises(FileNotFoundError, match="resnet-50 weights"):       # reference behaviour when the ImageNet weights are absent
        demo.update_config_files(demo.DEMO_PROJECT)
    demo.return_configs(demo.DEMO_PROJECT)
    proj = demo.update_config_files(demo.DEMO_PROJECT, need_init_weights=False)
    assert proj == str(dst)
    cfg = dcfg.read_config(str(dst / "config.yaml"))
    assert cfg["project_path"] == str(dst) and list(cfg["video_sets"]) == [str(dst / "videos" / "reachingvideo1.avi")]
    assert list(cfg["bodyparts"]) == BODYPARTS and [list(s) for s in cfg["skeleton"]] == SKELETON
    folder = dcfg.GetModelFolder(cfg["TrainingFraction"][0], 1, cfg)
    assert str(folder).endswith(os.path.join("iteration-0", "ReachingAug30-trainset95shuffle1"))
    for dtype in ("train", "test"):
        pc = dcfg.load_config(str(dst / folder / dtype / "pose_cfg.yaml"))
        assert pc.num_joints == 5 and pc.net_type == "resnet_50" and pc.stride == 8.0 and abs(pc.locref_stdev - 7.2801) < 1e-9
    snap, cfg_yaml = get_snapshot_path("snapshot-step0-final--0", proj, shuffle=1)
    assert snap.endswith(os.path.join("train", "snapshot-step0-final--0")) and os.path.samefile(cfg_yaml, dst / "config.yaml")

    # the video is absent (.MISSING_LARGE_BLOBS): its 55 labeled PNGs stand in, frame NNN = img<NNN>.png
    src = open_frame_source(str(dst / "videos" / "reachingvideo1.avi"))
    assert isinstance(src, LabeledDirSource) and len(src.files) == 55 and src.n_frames == 246 and src.size == (832, 747)
    from PIL import Image
    with Image.open(dst / "labeled-data" / "reachingvideo1" / "img023.png") as im:
        ref23 = np.asarray(im.convert("RGB"))
    assert np.array_equal(src.frame_at(23), ref23) and np.array_equal(src.frame_at(22), src.frame_at(20))
    assert sum(1 for _ in src.iter_frames()) == 246

    S0 = dcfg.skeleton_matrix(cfg)
    db = MultiDataset(config_yaml=str(dst / "config.yaml"), video_sets=_video_sets(dst, cfg), shuffle=1, S0=S0)
    d0 = db.datasets[0]
    assert (d0.nx_in, d0.ny_in, d0.nx_out, d0.ny_out) == (747, 832, 94, 104)            # SURVEY 8(a) B11 demo dims
    assert np.array_equal(d0.labels_idxs_all, np.sort(G["frame_idxs"]))
    db.create_batches_from_resnet_output(0, ns_jump=None, step=1, ns=10, nc=2048, n_max_frames=2000)
    assert db.n_visible_frames_total == 52 and db.n_frames_total >= 52 and db.nj == 5
    gcfg = db.dlc_config
    gcfg.update(ws=1000, ws_max=1.2, wt=0, wt_max=0, wn_visible=5, wn_hidden=3, gamma=1, gauss_len=1, lengthscale=1, batch_size=10,
                n_times_all_frames=100, lr=0.005, gm2=1, gm3=3, aug=True)
    loss, total_loss, total_loss_visible, placeholders = dgp_loss(db, gcfg)
    assert tuple(placeholders.keys()) == PLACEHOLDER_KEYS and len(placeholders) == 12
    np.testing.assert_allclose(loss.graph.ws, G["ws"], rtol=1e-12)
    np.testing.assert_allclose(loss.graph.ws_max, G["ws_max"], rtol=1e-12)
    sched = gen_batch([d.idxs["pv"] for d in db.datasets], [d.idxs["ph"] for d in db.datasets],
                      [d.idxs["chunk"] for d in db.datasets], gcfg, 5)
    (vis, hid, _, images, joint_loc, mask, _, addn), _ = db.next_batch(0, 0, np.array([5, 20]), np.array([6, 7, 21]))
    assert images.shape == (5, 747, 832, 3) and images.dtype == np.uint8 and joint_loc.shape == (2, 5, 2)
    assert len(sched) == 5 and list(mask) == [1, 1, 0, 1]

    demo.return_configs(demo.DEMO_PROJECT)
    after = {p: yaml.safe_load(open(p)) for p in before}
    for p, txt in before.items():
        assert after[p] == yaml.safe_load(txt), p                # the project is back to its shipped (relative-path) state
