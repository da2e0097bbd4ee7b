"""Host-side pieces added in round 2: augmentation of the labeled frames (B10 data_aug), the session-shaped training boundary
(dgp_loss return contract), the packed trajectory layout, the pose_net weight transform."""
import numpy as np
import pytest


def _dot_image(H, W, pts, r=2):
    img = np.zeros((H, W, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    for c, (x, y) in enumerate(pts):
        m = (yy - y) ** 2 + (xx - x) ** 2 <= r * r
        img[m, c % 3] = 255
    return img


def _centroid(img, ch):
    yy, xx = np.nonzero(img[..., ch] > 80)
    return np.array([xx.mean(), yy.mean()])


@pytest.mark.parametrize("name", ["fliplr", "rotate", "motion_blur", "coarse_dropout", "elastic", "gaussian_noise", "crop_and_pad"])
def test_each_augmenter_moves_keypoints_with_the_pixels(name):
    """A bright dot per keypoint: after the augmenter the dot's centroid is where the transformed keypoint says (<= 1.5 px)."""
    from deepgraphpose_amd.augment import NumpyAugPipeline
    H, W = 120, 160
    pts = np.array([[40.0, 30.0], [100.0, 70.0], [80.0, 95.0]])
    img = _dot_image(H, W, pts, r=3)
    for seed in range(6):
        p = NumpyAugPipeline(1.0, seed=seed)
        out, kp = getattr(p, name)(img.copy(), pts.copy())
        assert out.shape == img.shape and out.dtype == np.uint8 and kp.shape == pts.shape
        for c in range(3):
            if (out[..., c] > 80).sum() < 5:
                continue                                 # dot cropped / dropped away
            if not (0 <= kp[c, 0] < W and 0 <= kp[c, 1] < H):
                continue
            assert np.abs(_centroid(out, c) - kp[c]).max() < 1.6, (name, seed, c, _centroid(out, c), kp[c])


def test_data_aug_contract_and_nan_labels():
    """data_aug (fitdgp_util.py:439-451): only the visible frames change, labels go (row, col) cells <-> (x, y) px and back, NaN
    labels stay NaN, a pipeline that never fires is the identity."""
    from deepgraphpose_amd.augment import build_aug, data_aug

    class Cfg:
        stride = 8.0
    rng = np.random.RandomState(0)
    batch = rng.randint(0, 255, (5, 64, 96, 3)).astype(np.uint8)
    jl = np.array([[[3.0, 4.0], [5.0, 6.0], [np.nan, np.nan]], [[1.0, 9.0], [np.nan, np.nan], [6.5, 2.25]]])
    ident = build_aug(apply_prob=0.0, seed=0, backend="numpy")
    ident.crop_and_pad = lambda im, kp: (im, kp)         # the last stage has its own probability (0.4)
    out, jl2 = data_aug(batch, [1, 3], jl, ident, Cfg)
    assert out.dtype == batch.dtype and np.array_equal(out, batch)
    np.testing.assert_allclose(jl2, jl, equal_nan=True, atol=1e-12)
    pipe = build_aug(apply_prob=0.8, seed=3, backend="numpy")
    out, jl3 = data_aug(batch, [1, 3], jl, pipe, Cfg)
    assert np.array_equal(out[[0, 2, 4]], batch[[0, 2, 4]]) and not np.array_equal(out[[1, 3]], batch[[1, 3]])
    assert np.array_equal(np.isnan(jl3), np.isnan(jl)) and jl3.shape == jl.shape
    a = build_aug(0.8, seed=11, backend="numpy")(images=batch[:2], keypoints=[[(1.0, 2.0)], [(3.0, 4.0)]])
    b = build_aug(0.8, seed=11, backend="numpy")(images=batch[:2], keypoints=[[(1.0, 2.0)], [(3.0, 4.0)]])
    assert np.array_equal(a[0], b[0]) and a[1] == b[1]   # seeded -> reproducible


def test_dgp_loss_returns_the_reference_contract():
    """dgp_loss -> (loss, total_loss, total_loss_visible, placeholders) with the reference's 12 placeholder keys (fitdgp.py:1130-1144)
    and evaluable handles; feed_dict validation happens before any GPU work."""
    from deepgraphpose_amd.config import AttrDict
    from deepgraphpose_amd.models.fitdgp import dgp_loss, _feed
    from deepgraphpose_amd.models.session import LossTensor, Placeholder, TrainOp, TrainSession, PLACEHOLDER_KEYS

    class DS:
        labels = np.array([[[1.0, 2.0], [4.0, 6.0], [np.nan, np.nan]], [[2.0, 2.0], [5.0, 9.0], [7.0, 1.0]]])

    class DB:
        S0 = np.array([[1.0, -1.0, 0.0], [0.0, 1.0, -1.0]])
        nj, datasets, n_frames_total, n_visible_frames_total = 3, [DS()], 40, 2
    cfg = AttrDict(ws=1000, ws_max=1.2, wt=0, wt_max=0, wn_visible=5, wn_hidden=3, gamma=1, gauss_len=1, lengthscale=1, lr=0.005,
                   gm2=1, gm3=3, stride=8.0, locref_loss_weight=0.05, locref_huber_loss=True)
    loss, total_loss, total_loss_visible, placeholders = dgp_loss(DB(), cfg)
    assert tuple(placeholders) == PLACEHOLDER_KEYS == ("inputs", "targets", "locref_map", "locref_mask", "visible_marker_pl",
                                                       "hidden_marker_pl", "visible_marker_in_targets_pl", "wt_batch_mask_pl",
                                                       "vector_field_tf", "nt_batch_pl", "wt_batch_pl", "alpha_tf")
    assert all(isinstance(v, Placeholder) for v in placeholders.values())
    assert {"visible_loss_pred", "hidden_loss_pred", "visible_loss_locref", "total_loss"} <= set(loss)
    assert isinstance(total_loss, LossTensor) and total_loss is loss["total_loss"] and total_loss_visible.name == "total_loss_visible"
    g = loss.graph
    assert g.ws.shape == (2,) and g.ws_max.shape == (2,) and np.all(g.ws > 0)
    lr = Placeholder("learning_rate")
    op = g.minimize(total_loss, lr)
    assert isinstance(op, TrainOp) and op.clip_norm == 10.0 and op.momentum == 0.9          # fitdgp.py:709-712
    with pytest.raises(ValueError):
        g.minimize(loss["visible_loss_pred"], lr)
    bad = AttrDict(cfg, gm2=3)
    with pytest.raises(Exception, match="Not implemented"):
        dgp_loss(DB(), bad)
    fd = _feed(placeholders, np.zeros((3, 16, 16, 3), np.uint8), DS.labels[:1], np.zeros((3, 2, 2, 6)), np.zeros((3, 2, 2, 6)),
               (np.array([0, 1]), np.array([2, 3, 4, 5, 6, 7, 8]), np.array([0, 1])), np.array([1, 0]), None, 0, 2, 2, lr, 0.005)
    assert len(fd) == 13 and fd[placeholders["nt_batch_pl"]] == 3 and fd[placeholders["alpha_tf"]].shape == (2, 2, 2)
    sess = TrainSession(trainer=None, graph=g)
    del fd[placeholders["targets"]]
    with pytest.raises(KeyError, match="targets"):
        sess.run([loss, op], fd)


def test_packed_trajectory_layout_round_trip():
    """[T,nj,5] records (row, col, likelihood, iy, ix-as-int32-bits): what dgp_infer_packed writes and the all-gather moves."""
    import torch
    from deepgraphpose_amd import dist as ddist
    rng = np.random.RandomState(1)
    mu = torch.from_numpy(rng.rand(7, 4, 2).astype(np.float32) * 50)
    conf = torch.from_numpy(rng.rand(7, 4).astype(np.float32))
    idx = torch.from_numpy(rng.randint(0, 80, (7, 4, 2)).astype(np.int32))
    buf = ddist.pack_keypoints(mu, conf, idx)
    assert buf.shape == (7, 4, 5) and buf.dtype == torch.float32
    m2, c2, i2 = ddist.unpack_keypoints(buf)
    assert torch.equal(m2, mu) and torch.equal(c2, conf) and torch.equal(i2, idx)
    raw = buf.numpy().view(np.int32)
    assert np.array_equal(raw[..., 3:5], idx.numpy())


def test_deconv_phase_weights_reproduce_the_transposed_conv():
    """pose_net._deconv_as_phase_conv: the 3x3 / stride-2 SAME transposed conv as a 2x2 conv over four output phases."""
    import torch
    from deepgraphpose_amd.nnet.pose_net import _deconv_as_phase_conv
    from oracle import dgp_oracle as O
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 5, 6, 8)).astype(np.float32)
    w = rng.standard_normal((3, 3, 3, 8)).astype(np.float32)
    b = rng.standard_normal(3).astype(np.float32)
    ref = np.asarray(O.conv2d_transpose_same(x, w, b))
    wp = _deconv_as_phase_conv(w)
    xp = torch.nn.functional.pad(torch.from_numpy(x).permute(0, 3, 1, 2), (1, 0, 1, 0))
    y = torch.nn.functional.conv2d(xp, torch.from_numpy(wp).permute(3, 2, 0, 1)).permute(0, 2, 3, 1)
    y = y.reshape(2, 5, 6, 2, 2, 3).permute(0, 1, 3, 2, 4, 5).reshape(2, 10, 12, 3).numpy() + b
    assert np.abs(y - ref).max() < 1e-5


def test_prefetched_keeps_order_propagates_errors_and_stops(monkeypatch):
    """fitdgp._prefetched: items come in order from ONE producer thread (so every random draw happens in schedule order, with or
    without prefetching), a producer exception surfaces in the consumer at the right item, and abandoning the generator stops the
    producer."""
    import threading
    import time
    from deepgraphpose_amd.models.fitdgp import _prefetched
    rng = np.random.RandomState(5)
    made = []

    def make(i):
        made.append((i, threading.get_ident()))
        return i, float(rng.random_sample())
    got = list(_prefetched(make, 20, depth=3))
    ref_rng = np.random.RandomState(5)
    assert got == [(i, float(ref_rng.random_sample())) for i in range(20)]
    assert len({t for _, t in made}) == 1 and made[0][1] != threading.get_ident()
    assert list(_prefetched(lambda i: i * i, 5, depth=0)) == [0, 1, 4, 9, 16]          # inline mode

    def boom(i):
        if i == 3:
            raise ValueError("bad frame %d" % i)
        return i
    out = []
    with pytest.raises(ValueError, match="bad frame 3"):
        for v in _prefetched(boom, 10, depth=2):
            out.append(v)
    assert out == [0, 1, 2]
    count = [0]

    def slow(i):
        count[0] += 1
        time.sleep(0.01)
        return i
    gen = _prefetched(slow, 1000, depth=2)
    assert next(gen) == 0
    gen.close()                                   # consumer walks away: the producer must stop, not build 1000 items
    time.sleep(0.2)
    assert count[0] < 20


def test_bench_spawner_stops_the_other_workers_when_one_dies():
    """`python bench.py --gpus 2` (its own spawner): rank 1 exits with code 7 before joining the process group; rank 0 would wait in the
    rendezvous for ever.  The parent polls every child, stops rank 0 and exits with rank 1's code -- no GPU involved (the workers never
    get as far as the device)."""
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DGP_BENCH_FAULT_RANK="1", DGP_DIST_BACKEND="gloo", PYTHONPATH=root, DGP_BENCH_VISIBLE_GPUS="2")      # (no device here: opt in)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--no-cpu-baseline"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 7, (r.returncode, r.stderr[-500:])
    assert "rank 1 exited with code 7" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert time.monotonic() - t0 < 200


def test_bench_refuses_more_ranks_than_visible_devices():
    """`python bench.py --gpus 2` where fewer than two devices are visible (none in this container) is an error unless DGP_BENCH_VISIBLE_GPUS says
    the sharing is intended: an N-rank line measured on fewer devices is not an N-GPU line."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DGP_BENCH_VISIBLE_GPUS"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--no-cpu-baseline"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and "device(s) visible" in r.stderr


def test_readme_numbers_are_generated_from_the_committed_bench_lines():
    """README.md's table of numbers is the output of scripts/readme_numbers.py over profiles/r<latest>_bench_line*.json -- never typed."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "readme_numbers.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_scripts_index_is_up_to_date():
    """scripts/README.md is the output of scripts/index.py (one line per script from its docstring / leading comment): every script says what it is."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "index.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, "run `python scripts/index.py`"
    assert "(no description)" not in open(os.path.join(root, "scripts", "README.md")).read()


def test_resolve_tier_argument_environment_and_rejects(monkeypatch):
    """The one argument the entry points add to the reference's signatures (round 6): None -> DGP_EVAL_TIER -> the library default."""
    from deepgraphpose_amd.models import eval as E
    monkeypatch.delenv("DGP_EVAL_TIER", raising=False)
    assert E.resolve_tier(None) is None and E.resolve_tier("parity") == "parity" and E.resolve_tier("f16") == "f16"
    assert E.resolve_tier("BF16") == "f16" and E.resolve_tier("fp16") == "f16"        # (what configs[3] calls the tier)
    monkeypatch.setenv("DGP_EVAL_TIER", "f16")
    assert E.resolve_tier(None) == "f16" and E.resolve_tier("parity") == "parity"     # the argument wins
    monkeypatch.setenv("DGP_EVAL_TIER", "int8")
    with pytest.raises(ValueError):
        E.resolve_tier(None)
    import inspect
    for fn in (E.estimate_pose, E.setup_dgp_eval_graph, E.evaluate_dgp, E.plot_dgp):
        assert inspect.signature(fn).parameters["tier"].default is None
    # the reference's own parameters are untouched, in order (eval.py:147, :217-218, :656-657, :816-818)
    assert list(inspect.signature(E.estimate_pose).parameters)[:9] == ["proj_cfg_file", "dgp_model_file", "video_file", "output_dir", "shuffle",
                                                                          "save_pose", "save_str", "new_size", "crop_size"]
    assert list(inspect.signature(E.setup_dgp_eval_graph).parameters)[:5] == ["dlc_cfg", "dgp_model_file", "loc_ref", "gauss_len", "gamma"]


def test_setup_dgp_eval_graph_keeps_one_session_per_snapshot(tmp_path, monkeypatch):
    """The session kept between calls (no GPU needed: engines are built on first use): same snapshot files + same arguments -> the same
    session object; another tier, changed file contents or DGP_EVAL_SESSION_CACHE=0 -> another one; clear_session_cache() drops it."""
    from deepgraphpose_amd import weights_io
    from deepgraphpose_amd.models import eval as E
    from deepgraphpose_amd.synthetic import make_weights

    class Cfg(dict):
        __getattr__ = dict.get
    cfg = Cfg(net_type="resnet_50", num_joints=3)
    wts = make_weights(50, 3, False, seed=1)
    snap = weights_io.save_weights(str(tmp_path / "snapshot-step2-final--0"), wts)
    monkeypatch.delenv("DGP_EVAL_TIER", raising=False)
    monkeypatch.delenv("DGP_EVAL_SESSION_CACHE", raising=False)
    E.clear_session_cache()
    s1 = E.setup_dgp_eval_graph(cfg, snap)[0]
    assert E.setup_dgp_eval_graph(cfg, snap)[0] is s1
    s1.close()                                               # tf.Session.close(): the kept session survives it
    assert E.setup_dgp_eval_graph(cfg, snap)[0] is s1
    assert E.setup_dgp_eval_graph(cfg, snap, gamma=2)[0] is not s1                       # another graph argument
    s2 = E.setup_dgp_eval_graph(cfg, snap, tier="f16")[0]
    assert s2.tier == "f16" and E.setup_dgp_eval_graph(cfg, snap, tier="f16")[0] is s2
    w2 = dict(wts)
    w2["pose/part_pred/block4/biases"] = wts["pose/part_pred/block4/biases"] + 1.0
    weights_io.save_weights(snap, w2)                                                    # same path, other bytes
    s3 = E.setup_dgp_eval_graph(cfg, snap, tier="f16")[0]
    assert s3 is not s2 and float(s3.weights["pose/part_pred/block4/biases"][0]) == float(w2["pose/part_pred/block4/biases"][0])
    monkeypatch.setenv("DGP_EVAL_SESSION_CACHE", "0")
    assert E.setup_dgp_eval_graph(cfg, snap, tier="f16")[0] is not s3
    cfg101 = Cfg(net_type="resnet_101", num_joints=3)
    with pytest.raises(KeyError):
        E.setup_dgp_eval_graph(cfg101, snap)                                             # (the reference's resnet_50 -> 101 fallback relies on it)
    with pytest.raises(FileNotFoundError):
        E.setup_dgp_eval_graph(cfg, str(tmp_path / "nope"))
    E.clear_session_cache()
    assert not E._SESSION_CACHE


def test_bench_counts_visible_gpus_without_touching_hip(monkeypatch):
    """bench.py's spawner decides whether `--gpus N` fits from sysfs and the *_VISIBLE_DEVICES variables alone (the parent process must
    never initialise the GPU): list parsing as in the runtime (-1 ends a list), the smallest of the lists wins."""
    import bench
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    base = bench.visible_gpu_count()
    assert base >= 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2,3")
    assert bench.visible_gpu_count() == (min(base, 4) if base else 4)
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0,1,-1,3")
    assert bench.visible_gpu_count() == (min(base, 2) if base else 2)
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == 0
