"""Drop-in counterparts of deepgraphpose/models/fitdgp_util.py for the hot path.

  argmax_2d_from_cm      HIP soft-argmax kernel (fitdgp_util.py:342-402)
  find_nan_ind / find_hidden_markers / find_visible_markers / gen_batch   host numpy (fitdgp_util.py:77-202)
  combine_all_marker     device scatter (fitdgp_util.py:232-272)
  get_snapshot_path      project-layout helper (fitdgp_util.py:205-229)
"""
from __future__ import annotations

import random
from pathlib import Path

import numpy as np

_EMPTY = np.empty(0, dtype="int")


# --------------------------------------------------------------------------------------- device ops
def argmax_2d_from_cm(tensor, nj, gamma=1, gauss_len=2, th=None):
    """(T x nx_out x ny_out x nj) scoremap -> (spatial_soft_argmax [T,nj,2] as (row, col),
    normalised blurred softmax [T,nx_out,ny_out,nj]); same signature as fitdgp_util.py:342.

    `tensor` is a device fp32 torch tensor; the computation is the gfx950 `soft_argmax` kernel."""
    from .. import engine
    if tensor.dim() != 4:
        raise AssertionError("rank(tensor) == 4")                      # fitdgp_util.py:357
    if tensor.shape[-1] != nj:
        raise ValueError("last dimension (%d) != nj (%d)" % (tensor.shape[-1], nj))
    mu, _, _, pmap = engine.soft_argmax(tensor.contiguous(), float(gamma), int(gauss_len), want_pmap=True)
    if th is not None:                                                 # fitdgp_util.py:377-388 (no reference driver passes th)
        mu = engine.pmap_threshold(pmap, float(th))
    return mu, pmap


def combine_all_marker(targets_pred_hidden_marker, targets_visible_marker, hidden_marker_pl, visible_marker_pl,
                       nj, nt_batch_pl):
    """Scatter hidden (predicted) and visible (label) coordinates into [nt*nj, 2]; fitdgp_util.py:232-272.
    torch tensors on any device; duplicate indices accumulate like tf.scatter_nd."""
    import torch
    out = torch.zeros((int(nt_batch_pl) * nj, 2), dtype=targets_pred_hidden_marker.dtype,
                      device=targets_pred_hidden_marker.device)
    if hidden_marker_pl.numel():
        out.index_add_(0, hidden_marker_pl.long(), targets_pred_hidden_marker)
    if visible_marker_pl.numel():
        out.index_add_(0, visible_marker_pl.long(), targets_visible_marker.to(out.dtype))
    return out


# --------------------------------------------------------------------------------------- host index helpers
def find_nan_ind(target_ind, joint_loc):
    """Sorted marker ids (frame*nj + joint) of NaN joints in the visible frames; fitdgp_util.py:77-101."""
    if len(target_ind) == 0:
        return np.empty(0, dtype="int")
    nvisible, nj, _ = joint_loc.shape
    assert len(target_ind) == nvisible
    fi, ji = np.nonzero(np.isnan(joint_loc[:, :, 0]))
    return list(np.sort(nj * np.asarray(target_ind)[fi] + ji))


def _frame_markers(frames, nj):
    frames = np.asarray(frames)
    return np.sort((frames[:, None] * nj + np.arange(nj)[None, :]).ravel())


def find_hidden_markers(hidden_frame, nj, nan_ind):
    """All markers of hidden frames plus the NaN markers of visible frames; fitdgp_util.py:104-122."""
    if len(hidden_frame) == 0:
        return np.empty(0, dtype="int")
    return np.sort(list(_frame_markers(hidden_frame, nj)) + list(nan_ind))


def find_visible_markers(visible_frame, nj, nan_ind):
    """-> (all markers of visible frames, those not NaN); fitdgp_util.py:125-143."""
    if len(visible_frame) == 0:
        assert len(nan_ind) == 0
        return np.empty(0, dtype="int"), np.empty(0, dtype="int")
    v0 = _frame_markers(visible_frame, nj)
    return v0, np.sort(np.setdiff1d(v0, nan_ind))


def gen_batch(visible_frame_total, hidden_frame_total, all_frame_total, dgp_cfg, maxiters, verbose: bool = True):
    """Pre-computed list of frame windows, one per iteration; fitdgp_util.py:146-202.

    Uses the global numpy / python RNGs in the reference's call order, so equal seeds give equal
    schedules.  Each element: int32 [batch_size + 1], the last entry is the dataset index."""
    batch_size = dgp_cfg.batch_size
    n_frames_total = np.sum([len(v) for v in all_frame_total])
    n_datasets = len(all_frame_total)
    nepoch = np.min([int(n_frames_total * dgp_cfg.n_times_all_frames / batch_size), maxiters])
    if verbose:
        print("nepoch: ", nepoch)
        print("n_datasets: ", n_datasets)
    out = []
    for i in range(n_datasets):
        pool = np.unique(list(visible_frame_total[i]) + list(all_frame_total[i]) + list(hidden_frame_total[i]))
        bs = dgp_cfg.batch_size
        n_draw = max([1, int(nepoch / n_frames_total * len(pool))])
        if len(pool) < bs:
            start = np.random.randint(0, len(pool), size=n_draw)
            bs = 1
        else:
            start = np.random.randint(0, len(pool) - bs, size=n_draw)
        pos = (start.reshape(-1, 1) + np.arange(bs).reshape(1, -1)).astype(int)
        rows = pool[pos.reshape(-1)].reshape(-1, bs)
        rows = np.hstack((rows, i * np.ones((rows.shape[0], 1))))
        out += [r.astype(np.int32) for r in rows]
    return random.sample(out, len(out))


def learn_wt(all_data_batch):
    """Optical-flow magnitude fields between consecutive frames of a batch, [nt-1, H, W] (fitdgp_util.py:454-467).
    The flow itself is third-party host code (cv2 Farneback, untouched); without OpenCV this raises."""
    try:
        import cv2
    except ImportError as e:
        raise ImportError("the temporal clique (wt > 0) needs OpenCV's Farneback optical flow on the host "
                          "(cv2 is not installed); run with wt=0 or pass batch['vector_field'] yourself") from e
    fields = []
    for ff in range(all_data_batch.shape[0] - 1):
        prvs = cv2.cvtColor(all_data_batch[ff].astype(np.uint8), cv2.COLOR_BGR2GRAY)
        nxt = cv2.cvtColor(all_data_batch[ff + 1].astype(np.uint8), cv2.COLOR_BGR2GRAY)
        flow = cv2.calcOpticalFlowFarneback(prvs, nxt, None, 0.5, 3, 15, 3, 5, 1.2, 0)
        fields.append(np.abs(flow).sum(2))
    return np.array(fields)


# --------------------------------------------------------------------------------------- layout helper
def get_snapshot_path(snapshot, dlcpath, shuffle=1, trainingsetindex=0):
    """-> (snapshot_path, config_path); fitdgp_util.py:205-229."""
    from .. import config as dcfg
    base = Path(dlcpath)
    config_path = base / "config.yaml"
    cfg = dcfg.read_config(config_path)
    folder = dcfg.GetModelFolder(cfg["TrainingFraction"][trainingsetindex], shuffle, cfg)
    return str(base / folder / "train" / snapshot), config_path
