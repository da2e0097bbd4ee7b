"""CPU tests of the host-side mirror: index helpers, batch schedule, locref targets vs reference golden
vectors; config layout; C-ABI symbol table; export round trip; frame sharding over gloo (world size 2)."""
import ctypes as C
import os
import random
import re
import socket
from types import SimpleNamespace

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))


def test_marker_index_helpers_match_reference():
    from deepgraphpose_amd import dataset as D
    from deepgraphpose_amd.models import fitdgp_util as F
    for i in range(int(GOLD["mk_n"])):
        vis, hid, jl = GOLD["mk%d_vis" % i], GOLD["mk%d_hid" % i], GOLD["mk%d_joint" % i]
        vm, hm, vt = D.gen_idx_chunk(vis, hid, jl)
        np.testing.assert_array_equal(vm, GOLD["mk%d_visible_marker" % i])
        np.testing.assert_array_equal(hm, GOLD["mk%d_hidden_marker" % i])
        np.testing.assert_array_equal(vt, GOLD["mk%d_visible_in_targets" % i])
        nan_ind = F.find_nan_ind(vis, jl)
        np.testing.assert_array_equal(np.asarray(nan_ind, dtype=np.int64), GOLD["mk%d_nan_ind" % i])
        np.testing.assert_array_equal(F.find_hidden_markers(hid, jl.shape[1], nan_ind), GOLD["mk%d_fu_hidden" % i])
        if ("mk%d_fu_visible" % i) in GOLD.files:
            v0, v1 = F.find_visible_markers(vis, jl.shape[1], nan_ind)
            np.testing.assert_array_equal(v0, GOLD["mk%d_fu_visible0" % i])
            np.testing.assert_array_equal(v1, GOLD["mk%d_fu_visible" % i])
        if ("mk%d_pv_ts" % i) in GOLD.files:
            pv_ts, ph_ts = D.find_marker_index(vis, hid, jl)
            np.testing.assert_array_equal(pv_ts, GOLD["mk%d_pv_ts" % i])
            np.testing.assert_array_equal(ph_ts, GOLD["mk%d_ph_ts" % i])


def test_empty_inputs():
    from deepgraphpose_amd import dataset as D
    from deepgraphpose_amd.models import fitdgp_util as F
    e = np.empty(0, dtype=int)
    vm, hm, vt = D.gen_idx_chunk(e, e, np.zeros((0, 3, 2)))
    assert len(vm) == len(hm) == len(vt) == 0
    assert len(F.find_nan_ind(e, np.zeros((0, 3, 2)))) == 0
    assert len(F.find_hidden_markers(e, 3, [])) == 0
    assert [len(a) for a in F.find_visible_markers(e, 3, [])] == [0, 0]


def test_window_and_hidden_frame_selection_match_reference():
    from deepgraphpose_amd import dataset as D
    for i in range(int(GOLD["sel_n"])):
        ns, n_frames, nmax, nsj = [int(v) for v in GOLD["sel%d_args" % i]]
        pv, pvh = GOLD["sel%d_pv" % i], GOLD["sel%d_pvh_sorted" % i]
        np.testing.assert_array_equal(D.get_neighboring_window(pv, ns, n_frames), GOLD["sel%d_window" % i])
        ph = D.select_hidden_frames(ns, pv, pvh, n_frames, nmax, None if nsj < 0 else nsj)
        np.testing.assert_array_equal(ph, GOLD["sel%d_ph" % i])


def test_gen_batch_matches_reference_with_same_seeds():
    from deepgraphpose_amd.models import fitdgp_util as F
    for i in range(int(GOLD["gb_n"])):
        bs, ntimes, maxit = [int(v) for v in GOLD["gb%d_args" % i]]
        vis = [GOLD["gb%d_vis%d" % (i, d)] for d in range(2)]
        hid = [GOLD["gb%d_hid%d" % (i, d)] for d in range(2)]
        al = [GOLD["gb%d_all%d" % (i, d)] for d in range(2)]
        np.random.seed(100 + i)
        random.seed(200 + i)
        out = F.gen_batch(vis, hid, al, SimpleNamespace(batch_size=bs, n_times_all_frames=ntimes), maxit, verbose=False)
        assert len(out) == int(GOLD["gb%d_n" % i])
        np.testing.assert_array_equal([len(b) for b in out], GOLD["gb%d_lens" % i])
        np.testing.assert_array_equal(np.concatenate(out), GOLD["gb%d_flat" % i])
        assert all(b.dtype == np.int32 for b in out)


def test_locref_targets_match_reference():
    from deepgraphpose_amd import dataset as D
    for i in range(int(GOLD["tg_n"])):
        thr, nj, h, w = [int(v) for v in GOLD["tg%d_args" % i]]
        sc, lmap, lmask = D.compute_target_part_scoremap([GOLD["tg%d_joint_id" % i]], [GOLD["tg%d_coords" % i]],
                                                         (h, w), nj, thr)
        np.testing.assert_array_equal(sc.astype(np.uint8), GOLD["tg%d_scmap" % i])
        np.testing.assert_array_equal(lmask.astype(np.uint8), GOLD["tg%d_locref_mask" % i])
        np.testing.assert_array_equal(lmap, GOLD["tg%d_locref_map" % i])
        t, m = D.coord2map(GOLD["tg%d_c2m_joint" % i], h, w, nj, thr)
        np.testing.assert_array_equal(m.astype(np.uint8), GOLD["tg%d_c2m_mask" % i])
        np.testing.assert_array_equal(t, GOLD["tg%d_c2m_targets" % i])


# ------------------------------------------------------------------------ config / layout
def _mk_project(tmp_path, nj=3):
    import yaml
    proj = tmp_path / "proj"
    train = proj / "dlc-models" / "iteration-0" / "ReachAug30-trainset95shuffle1" / "train"
    train.mkdir(parents=True)
    parts = ["p%d" % i for i in range(nj)]
    cfg = dict(Task="Reach", date="Aug30", iteration=0, TrainingFraction=[0.95], bodyparts=parts,
               skeleton=[[parts[0], parts[1]]], project_path=str(proj), scorer="me", pcutoff=0.4)
    (proj / "config.yaml").write_text(yaml.safe_dump(cfg))
    (train / "pose_cfg.yaml").write_text(yaml.safe_dump(dict(num_joints=nj, all_joints_names=parts,
                                                             net_type="resnet_50", pos_dist_thresh=17,
                                                             location_refinement=True, locref_loss_weight=0.05)))
    return proj, cfg


def test_config_layout(tmp_path):
    from deepgraphpose_amd import config as K
    from deepgraphpose_amd.models.fitdgp_util import get_snapshot_path
    proj, cfg = _mk_project(tmp_path)
    assert str(K.GetModelFolder(0.95, 1, cfg)) == "dlc-models/iteration-0/ReachAug30-trainset95shuffle1"
    assert str(K.GetTrainingSetFolder(cfg)) == "training-datasets/iteration-0/UnaugmentedDataSet_ReachAug30"
    d = K.get_train_config(K.read_config(proj / "config.yaml"), shuffle=1)
    assert d.stride == 8.0 and d.locref_stdev == 7.2801 and d.num_joints == 3       # defaults merged
    assert d.pos_dist_thresh == 17 and d.snapshot_prefix.endswith("train/snapshot")
    sp, cp = get_snapshot_path("snapshot-step2-final--0", str(proj), shuffle=1)
    assert sp.endswith("ReachAug30-trainset95shuffle1/train/snapshot-step2-final--0") and cp.name == "config.yaml"
    with pytest.raises(FileNotFoundError):
        K.read_config(proj / "nope.yaml")
    with pytest.raises(FileNotFoundError):
        K.get_train_config(K.read_config(proj / "config.yaml"), shuffle=7)
    S0 = K.skeleton_matrix(cfg)
    assert S0.tolist() == [[1, -1, 0]]
    a, b = K.load_config(str(proj / "dlc-models/iteration-0/ReachAug30-trainset95shuffle1/train/pose_cfg.yaml")), None
    a.ws = 1000          # per-call objects: hyper-parameters do not leak through a global singleton
    b = K.load_config(str(proj / "dlc-models/iteration-0/ReachAug30-trainset95shuffle1/train/pose_cfg.yaml"))
    assert "ws" not in b


def test_export_pose_like_dlc_roundtrip(tmp_path):
    from deepgraphpose_amd.models import eval as E
    rng = np.random.default_rng(0)
    labels = {"x": rng.random((7, 3)) * 100, "y": rng.random((7, 3)) * 100, "likelihoods": rng.random((7, 3))}
    E.export_pose_like_dlc(labels, "snapshot-step2-final--0", ["a", "b", "c"], str(tmp_path / "vid_labeled"))
    txt = (tmp_path / "vid_labeled.csv").read_text().splitlines()
    assert txt[0].startswith("scorer,") and txt[1].startswith("bodyparts,a,a,a,b") and txt[2].startswith("coords,x,y,likelihood")
    back = E.load_pose_from_dlc_to_dict(str(tmp_path / "vid_labeled.csv"))
    for k in labels:
        np.testing.assert_allclose(back[k], labels[k], rtol=1e-12)


def test_weights_io_and_errors(tmp_path):
    from deepgraphpose_amd import weights_io as Wio
    from deepgraphpose_amd.synthetic import make_weights
    w = make_weights(50, 2, False, seed=1)
    p = Wio.save_weights(str(tmp_path / "snapshot-step2-final--0"), w)          # default: the reference's on-disk format (TF V2 bundle)
    assert os.path.isfile(p + ".index") and os.path.isfile(p + ".data-00000-of-00001") and Wio.exists(p)
    back = Wio.load_weights(str(tmp_path / "snapshot-step2-final--0"))
    assert set(back) == set(w) and Wio.net_depth(back) == 50
    q = Wio.save_weights(str(tmp_path / "opt-out"), w, fmt="npz")
    assert q.endswith(".npz") and set(Wio.load_weights(str(tmp_path / "opt-out"))) == set(w) and Wio.exists(str(tmp_path / "opt-out"))
    assert not Wio.exists(str(tmp_path / "missing"))
    np.testing.assert_array_equal(back["pose/part_pred/block4/weights"], w["pose/part_pred/block4/weights"])
    with pytest.raises(FileNotFoundError):
        Wio.load_weights(str(tmp_path / "missing"))
    (tmp_path / "tfsnap.index").write_bytes(b"x")
    with pytest.raises(ValueError):                              # not an SSTable -> loud, never silently empty
        Wio.load_weights(str(tmp_path / "tfsnap"))


def test_frame_sources(tmp_path):
    from PIL import Image
    from deepgraphpose_amd.frames import open_frame_source
    arr = (np.random.default_rng(0).random((3, 20, 30, 3)) * 255).astype(np.uint8)
    for i in range(3):
        Image.fromarray(arr[i]).save(tmp_path / ("img%03d.png" % i))
    src = open_frame_source(str(tmp_path))
    assert src.n_frames == 3 and src.size == (30, 20)
    np.testing.assert_array_equal(np.stack(list(src.iter_frames())), arr)
    np.save(tmp_path / "v.npy", arr)
    assert open_frame_source(str(tmp_path / "v.npy")).n_frames == 3
    with pytest.raises(FileNotFoundError):
        open_frame_source(str(tmp_path / "nope.avi"))


# ------------------------------------------------------------------------ C-ABI
def test_cabi_library_exports_every_declared_symbol(lib_built):
    from deepgraphpose_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "dgp_hip.h")).read()
    declared = set(re.findall(r"\b(dgp_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"dgp_net_desc", "dgp_tensor_view", "dgp_conv_desc"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = C.CDLL(lib_built)
    for name in declared:
        assert hasattr(lib, name), name
    l = _lib.load()
    assert l.dgp_version() == 1
    # host-only entry points work without a GPU
    assert l.dgp_packed_weight_floats(3, 3, 64, 64) == 18 * 8 * 64 * 4
    assert l.dgp_packed_weight_floats(1, 1, 6, 64) == 0          # Cin % 4 != 0 is rejected
    from deepgraphpose_amd import engine
    w = np.arange(1 * 1 * 8 * 4, dtype=np.float32).reshape(1, 1, 8, 4)
    pk = engine.pack_conv_weights(w).reshape(8, 32, 4)             # [chunks (padded to 8)][CoutP=32][4]
    assert pk[0, 2, 1] == w[0, 0, 1, 2] and pk[1, 3, 0] == w[0, 0, 4, 3] and pk[2:].sum() == 0
    with pytest.raises(_lib.DgpError):
        engine.pack_conv_weights(np.zeros((1, 1, 6, 4), np.float32))


def test_engine_fails_loudly_without_gpu(lib_built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from deepgraphpose_amd import engine, _lib
    with pytest.raises(_lib.DgpError):
        engine.DGPNet(50, 4, 96, 128)
    with pytest.raises(_lib.DgpError):
        engine.soft_argmax(torch.zeros(1, 4, 4, 1))


# ------------------------------------------------------------------------ multi-process sharding (gloo, world 2)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, T, q):
    import torch
    import torch.distributed as dist
    from deepgraphpose_amd import dist as dd
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dd.init_from_env("gloo")
    lo, hi = dd.shard_range(T, rank, world)
    nj = 3
    fr = torch.arange(lo, hi, dtype=torch.float32)
    mu = torch.stack([fr[:, None].expand(-1, nj) + 0.25, fr[:, None].expand(-1, nj) * 2], -1)
    conf = fr[:, None].expand(-1, nj) / 100
    idx = torch.stack([fr[:, None].expand(-1, nj), torch.arange(nj)[None, :].expand(hi - lo, -1)], -1).to(torch.int32)
    full = dd.gather_trajectory(dd.pack_keypoints(mu.contiguous(), conf.contiguous(), idx.contiguous()), T)
    m, c, i = dd.unpack_keypoints(full)
    q.put((rank, m.numpy(), c.numpy(), i.numpy()))
    dist.destroy_process_group()


@pytest.mark.parametrize("T", [10, 7])
def test_frame_sharding_allgather_gloo_world2(T):
    import torch.multiprocessing as mp
    from deepgraphpose_amd import dist as dd
    assert dd.shard_range(7, 0, 2) == (0, 4) and dd.shard_range(7, 1, 2) == (4, 7) and dd.shard_range(3, 3, 8) == (3, 3)
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, T, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, m, c, i in res:
        fr = np.arange(T, dtype=np.float32)
        np.testing.assert_array_equal(m[:, 0, 0], fr + 0.25)
        np.testing.assert_array_equal(m[:, 2, 1], fr * 2)
        np.testing.assert_array_equal(c[:, 1], fr / 100)
        np.testing.assert_array_equal(i[:, 0, 0], np.arange(T))             # int32 survives the bit-cast transport
        np.testing.assert_array_equal(i[:, 2, 1], np.full(T, 2))


@pytest.mark.parametrize("T", [13, 5])
def test_frame_sharding_allgather_gloo_world8(T):
    """Eight ranks, T % 8 != 0: ceil(T / 8) frames per rank, a short last shard (T = 13: ranks 6 holds one frame, rank 7 none) or several
    EMPTY shards (T = 5: ranks 5-7) -- every rank still ends with the whole frame-ordered trajectory."""
    import torch.multiprocessing as mp
    from deepgraphpose_amd import dist as dd
    assert [dd.shard_range(13, r, 8) for r in range(8)] == [(0, 2), (2, 4), (4, 6), (6, 8), (8, 10), (10, 12), (12, 13), (13, 13)]
    assert [dd.shard_range(5, r, 8) for r in range(8)] == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 5), (5, 5), (5, 5)]
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 8, port, T, q)) for r in range(8)]
    for p in ps:
        p.start()
    res = [q.get(timeout=240) for _ in ps]
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(8))
    for rank, m, c, i in res:
        fr = np.arange(T, dtype=np.float32)
        assert m.shape == (T, 3, 2)
        np.testing.assert_array_equal(m[:, 0, 0], fr + 0.25)
        np.testing.assert_array_equal(m[:, 2, 1], fr * 2)
        np.testing.assert_array_equal(c[:, 1], fr / 100)
        np.testing.assert_array_equal(i[:, 0, 0], np.arange(T))
        np.testing.assert_array_equal(i[:, 2, 1], np.full(T, 2))


def _grad_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from deepgraphpose_amd import dist as dd
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dd.init_from_env("gloo")
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    dd.average_gradients(g, bucket_floats=256)                  # 4 buckets
    q.put((rank, g.numpy()))
    dist.destroy_process_group()


def test_gradient_averaging_gloo_world2():
    """N4: bucketed all-reduce of the flat gradient buffer = the mean over ranks, identical on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    ps = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, g in res:
        np.testing.assert_array_equal(g, np.arange(1000, dtype=np.float32) * 1.5)


def test_dlc_pose_dataset_samples(tmp_path):
    """Step-0 loader (pose_defaultdataset.PoseDataset): seeded schedule is reproducible, target maps have the
    network's scoremap size for every jittered / cropped frame size, disks sit on the scaled labels."""
    from _project import make_project
    from deepgraphpose_amd import config as K
    from deepgraphpose_amd.dlc_dataset import LearningRate, PoseDataset
    from deepgraphpose_amd.arch import scoremap_hw
    proj, frames, _ = make_project(tmp_path, hw=(120, 160))
    cfg = K.get_train_config(K.read_config(os.path.join(proj, "config.yaml")), shuffle=1)
    cfg.crop, cfg.cropratio, cfg.global_scale, cfg.pos_dist_thresh = True, 0.5, 0.8, 8
    cfg.minsize, cfg.leftwidth, cfg.rightwidth, cfg.topheight, cfg.bottomheight = 30, 40, 40, 40, 40

    def run(seed, n):
        np.random.seed(seed); random.seed(seed)
        ds = PoseDataset(cfg)
        return [ds.next_batch() for _ in range(n)]

    # skip_batch() consumes exactly the draws of next_batch(): rank 1 of a 3-rank job sees samples 1, 4, 7 of the common sequence
    np.random.seed(5); random.seed(5)
    ds = PoseDataset(cfg)
    seq = [ds.next_batch() for _ in range(9)]
    np.random.seed(5); random.seed(5)
    ds = PoseDataset(cfg)
    for it in range(3):
        got = None
        for r in range(3):
            if r == 1:
                got = ds.next_batch()
            else:
                ds.skip_batch()
        np.testing.assert_array_equal(got["inputs"], seq[3 * it + 1]["inputs"])
        np.testing.assert_array_equal(got["locref_targets"], seq[3 * it + 1]["locref_targets"])
    a, b = run(4, 9), run(4, 9)
    sizes = set()
    for x, y in zip(a, b):
        assert x["data_item"].im_path == y["data_item"].im_path
        np.testing.assert_array_equal(x["inputs"], y["inputs"])
        np.testing.assert_array_equal(x["part_score_targets"], y["part_score_targets"])
        _, H, W, _ = x["inputs"].shape
        sizes.add((H, W))
        assert x["inputs"].dtype == np.uint8
        assert x["part_score_targets"].shape == (1,) + tuple(scoremap_hw(H, W)) + (3,)
        assert x["locref_mask"].shape == (1,) + tuple(scoremap_hw(H, W)) + (6,)
        assert (x["part_score_weights"] == 1).all()
        np.testing.assert_array_equal(x["locref_mask"][..., 0::2], x["part_score_targets"])
    assert len(sizes) > 3                                          # scale jitter and crops really change the size
    assert len({x["data_item"].im_path for x in a[:4]}) == 4       # one pass visits every labeled image once
    lr = LearningRate(SimpleNamespace(multi_step=[[0.001, 2], [0.005, 4]]))
    assert [lr.get_lr(i) for i in range(5)] == [0.001, 0.001, 0.001, 0.005, 0.005]


def test_motion_energy_host_backend_equals_oracle_restatement():
    """calculate_motion_energy (DGP/dataset.py:29-43): uint8 differences wrap; frame 0 has no predecessor."""
    from deepgraphpose_amd import dataset as D
    from deepgraphpose_amd.frames import ArraySource
    from oracle import dgp_oracle as O
    rng = np.random.default_rng(3)
    clip = rng.integers(0, 256, (9, 6, 7, 3), dtype=np.uint8)
    clip[5] = clip[4]
    me = D.calculate_motion_energy(ArraySource(clip), backend="host")
    assert me[0] == 0.0 and me[5] == 0.0
    assert np.array_equal(me, O.motion_energy(clip))
    # the wrap: 10 - 250 = 16 (mod 256), not 240
    two = np.stack([np.full((2, 2, 3), 250, np.uint8), np.full((2, 2, 3), 10, np.uint8)])
    assert D.calculate_motion_energy(ArraySource(two), backend="host")[1] == 16.0
    with pytest.raises(ValueError):
        D.calculate_motion_energy(ArraySource(two), backend="numpy")


def test_cpulist_parser_and_numa_binding_is_a_noop_without_sysfs(monkeypatch):
    """dist.parse_cpulist reads sysfs' cpulist syntax; bind_to_gpu_numa_node never widens an affinity mask and does nothing when the GPU's
    NUMA node is unknown (-1) or DGP_NUMA_BIND=0."""
    from deepgraphpose_amd import dist as D
    assert D.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert D.parse_cpulist("") == [] and D.parse_cpulist("5") == [5]
    before = os.sched_getaffinity(0)
    monkeypatch.setattr(D, "device_identity", lambda lr: {"index": lr, "name": "x", "pci": "0000:00:00.0", "uuid": "", "numa_node": -1})
    assert D.bind_to_gpu_numa_node(0) == {"numa_node": -1, "cpus_bound": None}
    monkeypatch.setattr(D, "device_identity", lambda lr: {"index": lr, "name": "x", "pci": "0000:00:00.0", "uuid": "", "numa_node": 0})
    monkeypatch.setenv("DGP_NUMA_BIND", "0")
    assert D.bind_to_gpu_numa_node(0)["cpus_bound"] is None
    monkeypatch.delenv("DGP_NUMA_BIND")
    got = D.bind_to_gpu_numa_node(0)                       # node 0 exists on this host or it does not: either way the mask never grows
    assert os.sched_getaffinity(0) <= before
    assert got["cpus_bound"] is None or got["cpus_bound"] == len(os.sched_getaffinity(0))
    os.sched_setaffinity(0, before)
