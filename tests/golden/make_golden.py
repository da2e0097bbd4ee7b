#!/usr/bin/env python3
"""Generate tests/golden/reference_vectors.npz by RUNNING the reference's own pure-numpy code.

Runs only in the build container (needs /root/reference; nothing here travels to the GPU box
except the .npz it writes).  TensorFlow, slim, h5py, moviepy, skimage, imgaug, cv2, easydict and
the deeplabcut package are absent, so permissive stub modules are registered first; only functions
whose bodies are pure numpy are then called.  numpy-2 removed np.int / np.asscalar, which the
reference uses (fitdgp_util.py:194, pose_defaultdataset.py:235-236) -> shimmed.

Pinned reference functions (paths under /root/reference/src):
  A4  deepgraphpose/models/eval.py:331-343        likelihood window (inline loop body, exec'd from the file)
  A6  DeepLabCut/.../nnet/predict.py:62-77        argmax_pose_predict
  B10 deepgraphpose/models/fitdgp_util.py:77-202  find_nan_ind, find_hidden_markers, find_visible_markers, gen_batch
  B10 deepgraphpose/dataset.py:46-271             select_hidden_frames, get_neighboring_window, find_marker_index,
                                                  gen_idx_chunk, coord2map
  B10 DeepLabCut/.../dataset/pose_defaultdataset.py:220-266   compute_target_part_scoremap
"""
import importlib.util
import os
import random
import sys
import textwrap
import types

import numpy as np

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.npz")


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        m = _Stub(self.__name__ + "." + name)
        setattr(self, name, m)
        return m

    def __call__(self, *a, **k):
        return _Stub("call")


def install_stubs():
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "asscalar"):
        np.asscalar = lambda a: a.item()
    names = ["tensorflow", "tensorflow.compat", "tensorflow.compat.v1", "tensorflow.contrib",
             "tensorflow.contrib.slim", "tensorflow.contrib.slim.nets", "tensorflow.contrib.slim.nets.resnet_v1",
             "tensorflow.python", "tensorflow.python.util", "tensorflow.python.util.deprecation", "h5py", "moviepy",
             "moviepy.editor", "skimage", "skimage.util", "skimage.draw", "imgaug", "imgaug.augmenters",
             "imgaug.augmentables", "cv2", "easydict", "deeplabcut", "deeplabcut.utils",
             "deeplabcut.utils.auxiliaryfunctions", "deeplabcut.utils.auxfun_videos",
             "deeplabcut.pose_estimation_tensorflow", "deeplabcut.pose_estimation_tensorflow.nnet",
             "deeplabcut.pose_estimation_tensorflow.nnet.net_factory",
             "deeplabcut.pose_estimation_tensorflow.dataset",
             "deeplabcut.pose_estimation_tensorflow.dataset.pose_dataset", "matplotlib", "matplotlib.pyplot", "PIL"]
    for n in names:
        if n not in sys.modules:
            sys.modules[n] = _Stub(n)
    sys.modules["tensorflow"].__version__ = "1.15.0"


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def likelihood_from_eval_py(scmap_np, mu_n_batch):
    """exec the loop body of estimate_pose (eval.py:329-343) on one frame."""
    src = open(os.path.join(REF, "deepgraphpose/models/eval.py")).read().split("\n")
    body = "\n".join(src[328:343])          # lines 329..343 (1-based)
    assert "markers[ii] = mu_n_batch[0]" in body and "likelihoods[ii, jj_idx]" in body, "eval.py layout changed"
    nj = scmap_np.shape[-1]
    ns = dict(np=np, nj=nj, ii=0, offset_mu_jj=0, mu_n_batch=mu_n_batch, scmap_np=scmap_np,
              markers=np.zeros((1, nj, 2)), mu_likelihoods=np.zeros((1, nj, 2)).astype("int"),
              likelihoods=np.zeros((1, nj)))
    exec(textwrap.dedent(body), ns)
    return ns["mu_likelihoods"][0].copy(), ns["likelihoods"][0].copy()


def main():
    install_stubs()
    g = {}
    rng = np.random.default_rng(20260101)

    # ---------------- A6 argmax_pose_predict
    predict = load(os.path.join(REF, "DeepLabCut/deeplabcut/pose_estimation_tensorflow/nnet/predict.py"), "ref_predict")
    cases = [(12, 16, 3, True), (12, 16, 3, False), (60, 80, 4, True), (47, 52, 5, True), (5, 7, 1, False)]
    g["app_n"] = len(cases)
    for i, (h, w, c, with_off) in enumerate(cases):
        logits = rng.standard_normal((h, w, c)).astype(np.float32) * 3
        if i == 0:
            logits[2, 3, 0] = logits[9, 1, 0] = 40.0        # saturated tie -> first index
        scmap = (1.0 / (1.0 + np.exp(-logits))).astype(np.float32)
        off = (rng.standard_normal((h, w, c, 2)).astype(np.float32) * np.float32(7.2801)) if with_off else None
        pose = predict.argmax_pose_predict(scmap, off, 8.0)
        g["app%d_logits" % i] = logits
        g["app%d_scmap" % i] = scmap
        if with_off:
            g["app%d_off" % i] = off
        g["app%d_pose" % i] = np.asarray(pose, dtype=np.float64)

    # ---------------- A4 likelihood window
    lk = [(12, 16, 3), (60, 80, 4), (9, 9, 2)]
    g["lik_n"] = len(lk)
    for i, (h, w, c) in enumerate(lk):
        s = (rng.standard_normal((1, h, w, c)) * 4).astype(np.float32)
        mu = np.stack([rng.uniform(0, h - 1, c), rng.uniform(0, w - 1, c)], 1).astype(np.float32)[None]
        if i == 2:
            mu[0, 0] = [8.0, 8.0]        # exactly on the last cell: window clipped by the array end
            mu[0, 1] = [0.0, 3.0]        # integer-valued: floor == ceil -> 1-wide window
        idx, lik = likelihood_from_eval_py(s, mu)
        g["lik%d_scmap" % i], g["lik%d_mu" % i], g["lik%d_idx" % i], g["lik%d_lik" % i] = s[0], mu[0], idx, lik

    # ---------------- B10 fitdgp_util helpers
    fu = load(os.path.join(REF, "deepgraphpose/models/fitdgp_util.py"), "ref_fitdgp_util")
    ds = load(os.path.join(REF, "deepgraphpose/dataset.py"), "ref_dataset")
    mk = []
    for i in range(8):
        nv, nh, nj = int(rng.integers(0, 5)), int(rng.integers(0, 6)), int(rng.integers(1, 6))
        frames = rng.permutation(12)[: nv + nh]
        vis, hid = np.sort(frames[:nv]), np.sort(frames[nv:])
        jl = rng.uniform(0, 50, size=(nv, nj, 2))
        nan_mask = rng.random((nv, nj)) < 0.3
        jl[nan_mask] = np.nan
        mk.append((vis, hid, jl))
    g["mk_n"] = len(mk)
    for i, (vis, hid, jl) in enumerate(mk):
        vm, hm, vt = ds.gen_idx_chunk(vis, hid, jl)
        g["mk%d_vis" % i], g["mk%d_hid" % i], g["mk%d_joint" % i] = vis, hid, jl
        g["mk%d_visible_marker" % i] = np.asarray(vm, dtype=np.int64)
        g["mk%d_hidden_marker" % i] = np.asarray(hm, dtype=np.int64)
        g["mk%d_visible_in_targets" % i] = np.asarray(vt, dtype=np.int64)
        nan_ind = fu.find_nan_ind(vis, jl)
        g["mk%d_nan_ind" % i] = np.asarray(nan_ind, dtype=np.int64)
        g["mk%d_fu_hidden" % i] = np.asarray(fu.find_hidden_markers(hid, jl.shape[1], nan_ind), dtype=np.int64)
        if len(vis) > 0 or len(nan_ind) == 0:
            v0, v1 = fu.find_visible_markers(vis, jl.shape[1], nan_ind)
            g["mk%d_fu_visible0" % i] = np.asarray(v0, dtype=np.int64)
            g["mk%d_fu_visible" % i] = np.asarray(v1, dtype=np.int64)
        if len(vis) > 0 and len(hid) > 0:
            pv_ts, ph_ts = ds.find_marker_index(vis, hid, jl)
            g["mk%d_pv_ts" % i] = np.asarray(pv_ts, dtype=np.int64)
            g["mk%d_ph_ts" % i] = np.asarray(ph_ts, dtype=np.int64)

    # windows / hidden-frame selection
    sel = []
    for i in range(4):
        n_frames = int(rng.integers(200, 500))
        pv = np.sort(rng.choice(n_frames, size=int(rng.integers(3, 12)), replace=False))
        me = rng.random(n_frames)
        pvh_sorted = np.argsort(-me)
        ns = int(rng.integers(2, 8))
        nmax = int(rng.integers(60, 200))
        ns_jump = [None, 0, 2, None][i]
        win = ds.get_neighboring_window(pv, ns, n_frames)
        ph = ds.select_hidden_frames(ns, pv, pvh_sorted, n_frames, nmax, ns_jump)
        sel.append(1)
        g["sel%d_args" % i] = np.array([ns, n_frames, nmax, -1 if ns_jump is None else ns_jump], dtype=np.int64)
        g["sel%d_pv" % i], g["sel%d_pvh_sorted" % i] = pv, pvh_sorted
        g["sel%d_window" % i], g["sel%d_ph" % i] = win, np.asarray(ph, dtype=np.int64)
    g["sel_n"] = len(sel)

    # gen_batch with pinned seeds
    from types import SimpleNamespace
    gb = []
    for i, (bs, ntimes, maxit) in enumerate([(10, 100, 50), (4, 3, 1000), (10, 100, 7)]):
        vis_tot = [np.sort(rng.choice(80, 6, replace=False)), np.sort(rng.choice(60, 4, replace=False))]
        hid_tot = [np.sort(rng.choice(80, 9, replace=False)), np.sort(rng.choice(60, 3, replace=False))]
        all_tot = [np.arange(10, 40), np.arange(0, 3) if i == 1 else np.arange(5, 25)]
        cfg = SimpleNamespace(batch_size=bs, n_times_all_frames=ntimes)
        np.random.seed(100 + i)
        random.seed(200 + i)
        out = fu.gen_batch(vis_tot, hid_tot, all_tot, cfg, maxit)
        g["gb%d_args" % i] = np.array([bs, ntimes, maxit], dtype=np.int64)
        for d in range(2):
            g["gb%d_vis%d" % (i, d)], g["gb%d_hid%d" % (i, d)], g["gb%d_all%d" % (i, d)] = vis_tot[d], hid_tot[d], all_tot[d]
        g["gb%d_n" % i] = len(out)
        lens = np.array([len(b) for b in out], dtype=np.int64)
        g["gb%d_lens" % i] = lens
        g["gb%d_flat" % i] = np.concatenate([np.asarray(b, dtype=np.int64) for b in out]) if len(out) else np.zeros(0, np.int64)
        gb.append(1)
    g["gb_n"] = len(gb)

    # ---------------- locref targets: compute_target_part_scoremap + coord2map
    pdd = load(os.path.join(REF, "DeepLabCut/deeplabcut/pose_estimation_tensorflow/dataset/pose_defaultdataset.py"),
               "ref_pose_defaultdataset")
    tgt = []
    for i, (thr, nj, size) in enumerate([(8, 3, (20, 24)), (17, 5, (47, 52)), (8, 4, (60, 80))]):
        pdata = object.__new__(pdd.PoseDataset)
        pdata.cfg = SimpleNamespace(pos_dist_thresh=thr, num_joints=nj, weigh_only_present_joints=False)
        pdata.stride, pdata.half_stride, pdata.locref_scale = 8.0, 4.0, 1.0 / 7.2801
        present = np.sort(rng.choice(nj, size=max(1, nj - 1), replace=False))
        coords = np.stack([rng.uniform(0, size[1] * 8, len(present)), rng.uniform(0, size[0] * 8, len(present))], 1)
        scmap, weights, lmap, lmask = pdata.compute_target_part_scoremap([present], [coords], 0, np.array(size), 1)
        g["tg%d_args" % i] = np.array([thr, nj, size[0], size[1]], dtype=np.int64)
        g["tg%d_joint_id" % i], g["tg%d_coords" % i] = present, coords
        g["tg%d_scmap" % i] = scmap.astype(np.uint8)
        g["tg%d_locref_map" % i], g["tg%d_locref_mask" % i] = lmap.astype(np.float64), lmask.astype(np.uint8)
        # coord2map (dataset.py:246-271): joint_loc in scoremap units (row, col), NaN for missing
        jl = np.stack([rng.uniform(0, size[0] - 1, (2, nj)), rng.uniform(0, size[1] - 1, (2, nj))], -1)
        jl[0, 0] = np.nan
        lt, lm = ds.coord2map(pdata, jl, size[0], size[1], nj)
        g["tg%d_c2m_joint" % i] = jl
        g["tg%d_c2m_targets" % i], g["tg%d_c2m_mask" % i] = lt.astype(np.float64), lm.astype(np.uint8)
        tgt.append(1)
    g["tg_n"] = len(tgt)

    # ---------------- A4 again, with mu = the soft-argmax OF the scoremap (round 4): the GPU kernel computes its own mu, so these cases
    # let its window indices / likelihoods be compared with the reference's lines directly.  mu comes from this repository's oracle
    # (argmax_2d_from_cm is TF code in the reference and cannot run); the eval.py loop body that turns (scmap, mu) into (idx, lik) is
    # the reference's own.  Own generator: the arrays above stay byte-identical.
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import dgp_oracle as O
    rng2 = np.random.default_rng(20261003)
    lm = [(60, 80, 4, 1), (47, 52, 5, 1), (24, 32, 3, 2), (90, 160, 20, 1)]
    g["likmu_n"] = len(lm)
    for i, (h, w, c, gl) in enumerate(lm):
        while True:                                              # redraw until no coordinate sits within 1e-3 of an integer (the window
            s = (rng2.standard_normal((1, h, w, c)) * 1.5).astype(np.float32)      # would then depend on the last bits of mu)
            yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
            for j in range(c):                                   # one or two blobs per map: mu between them, windows off the peak too
                for _ in range(1 + (j % 2)):
                    py, px = rng2.uniform(2, h - 3), rng2.uniform(2, w - 3)
                    s[0, :, :, j] += (12.0 * np.exp(-((yy - py) ** 2 + (xx - px) ** 2) / (2 * rng2.uniform(0.8, 2.5) ** 2))).astype(np.float32)
            mu, _ = O.argmax_2d_from_cm(s, 1.0, gl)
            if np.abs(mu - np.round(mu)).min() > 1e-3:
                break
        idx, lik = likelihood_from_eval_py(s, mu)
        g["likmu%d_scmap" % i], g["likmu%d_mu" % i], g["likmu%d_idx" % i], g["likmu%d_lik" % i] = s[0], mu[0], idx, lik
        g["likmu%d_gauss_len" % i] = gl

    np.savez_compressed(OUT, **g)
    print("wrote", OUT, "%.1f KiB" % (os.path.getsize(OUT) / 1024), len(g), "arrays")


if __name__ == "__main__":
    main()
