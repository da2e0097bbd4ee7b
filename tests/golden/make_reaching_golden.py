#!/usr/bin/env python3
"""Generate tests/golden/reaching_vectors.npz (SURVEY.md 8(c) fixture (5)) by RUNNING reference code on the demo project that
ships with the reference (data/Reaching-Mackenzie-2018-08-30: the .mat training set, config.yaml; 57 labeled PNGs).

Build container only (needs /root/reference); only the .npz travels.  What runs from the reference, under the stub modules of
make_golden.py:
  * PET/dataset/pose_defaultdataset.py:39-76      PoseDataset.load_dataset on Reaching_Mackenzie95shuffle1.mat (real scipy.io)
  * DGP/dataset.py (the `targets_2d` loop of Dataset._compute_targets, :641-652), exec'd from the file on the loaded joints
  * DGP/models/fitdgp.py:607-617 (S0 from the skeleton) and :875-892 (limb statistics -> ws, ws_max), exec'd from the file
    (both sit inside TF-importing functions: the lines are cut out and run on numpy inputs, nothing is restated here)
"""
import os
import sys
import textwrap

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, install_stubs, load      # noqa: E402

PROJ = "/root/reference/data/Reaching-Mackenzie-2018-08-30"
OUT = os.path.join(HERE, "reaching_vectors.npz")


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def cut(path, first_marker, last_marker, include_last=True):
    """Lines of `path` from the first line containing first_marker to the (first following) line containing last_marker."""
    src = open(path).read().split("\n")
    a = next(i for i, l in enumerate(src) if first_marker in l)
    b = next(i for i in range(a, len(src)) if last_marker in src[i])
    return textwrap.dedent("\n".join(src[a:b + (1 if include_last else 0)]))


def main():
    install_stubs()
    g = {}
    proj = yaml.safe_load(open(os.path.join(PROJ, "config.yaml")))
    train_cfg = yaml.safe_load(open(os.path.join(
        PROJ, "dlc-models/iteration-0/ReachingAug30-trainset95shuffle1/train/pose_cfg.yaml")))
    nj, stride = int(train_cfg["num_joints"]), 8.0

    # ---- the reference loader on the shipped .mat
    pdd = load(os.path.join(REF, "DeepLabCut/deeplabcut/pose_estimation_tensorflow/dataset/pose_defaultdataset.py"),
               "ref_pose_defaultdataset")
    pd_ = object.__new__(pdd.PoseDataset)
    pd_.cfg = AttrDict(project_path=PROJ, dataset=train_cfg["dataset"], num_joints=nj)
    data = pd_.load_dataset()
    g["n_items"] = len(data)
    g["im_paths"] = np.array([str(d.im_path) for d in data])
    g["im_sizes"] = np.array([np.asarray(d.im_size).ravel() for d in data], dtype=np.int64)
    g["joints_len"] = np.array([d.joints[0].shape[0] for d in data], dtype=np.int64)
    g["joints_flat"] = np.concatenate([np.asarray(d.joints[0], dtype=np.float64) for d in data])      # rows (joint id, x, y)

    # ---- Dataset._compute_targets: frame selection as :616-637 (items of this video, first occurrence), then the file's own loop
    video = "reachingvideo1"
    frame_idxs, joinss = [], []
    for d in data:
        parts = os.path.normpath(str(d.im_path)).split(os.sep)
        if video not in parts:
            continue
        idx = int(os.path.split(str(d.im_path))[-1][3:].split(".")[0])
        if idx in frame_idxs:
            continue
        frame_idxs.append(idx)
        joinss.append(np.copy(d.joints[0]))
    nt = len(frame_idxs)
    body = cut(os.path.join(REF, "deepgraphpose/dataset.py"), "targets_2d = np.zeros((nt, nj, 2)) * np.nan",
               "(joinss_ntt_njj - stride / 2) / stride)")
    ns = dict(np=np, nt=nt, nj=nj, joinss=joinss, stride=stride)
    exec(body, ns)
    targets_2d = ns["targets_2d"]
    g["frame_idxs"], g["targets_2d"] = np.asarray(frame_idxs, dtype=np.int64), targets_2d

    # ---- S0 (fitdgp.py:607-617) and limb statistics (fitdgp.py:875-892), both exec'd from the file
    fit = os.path.join(REF, "deepgraphpose/models/fitdgp.py")
    body = cut(fit, "bodyparts = cfg['bodyparts']", "S0[s, skj] = -1")
    ns = dict(np=np, cfg=proj)
    exec(body, ns)
    S0 = ns["S0"]
    g["S0"] = S0
    body = cut(fit, "joint_locs = [d.labels for d in data_batcher.datasets]", "limb_full) + 1e-20) * dgp_cfg.ws")
    ds = AttrDict(labels=targets_2d)
    ns = dict(np=np, data_batcher=AttrDict(datasets=[ds]), nj=nj, S0=S0, dgp_cfg=AttrDict(stride=stride, ws=1000, ws_max=1.2))
    exec(body, ns)
    g["ws"], g["ws_max"] = np.asarray(ns["ws"], dtype=np.float64), np.asarray(ns["ws_max"], dtype=np.float64)
    # a second skeleton (dense, all pairs of the 5 bodyparts) through the same lines
    pairs = [(a, b) for a in range(nj) for b in range(a + 1, nj)]
    S1 = np.zeros((len(pairs), nj))
    for l, (a, b) in enumerate(pairs):
        S1[l, a], S1[l, b] = 1, -1
    ns = dict(np=np, data_batcher=AttrDict(datasets=[ds]), nj=nj, S0=S1, dgp_cfg=AttrDict(stride=stride, ws=1000, ws_max=1.2))
    exec(body, ns)
    g["dense_S0"], g["dense_ws"], g["dense_ws_max"] = S1, np.asarray(ns["ws"]), np.asarray(ns["ws_max"])
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, "%.1f KiB" % (os.path.getsize(OUT) / 1024), "items", len(data), "frames", nt, "ws", g["ws"], "ws_max", g["ws_max"])


if __name__ == "__main__":
    main()
