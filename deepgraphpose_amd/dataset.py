"""Host-side index / target bookkeeping of the DGP training batch.

Restates the caller side of the training step -- which markers of a batch are visible / hidden,
which frames around labeled frames are worth training on, and the locref target maps -- with the
reference's exact outputs (pinned by tests/golden/reference_vectors.npz):

  gen_idx_chunk, find_marker_index      DGP/dataset.py:157-239
  get_neighboring_window                DGP/dataset.py:103-119
  select_hidden_frames                  DGP/dataset.py:46-101
  compute_target_part_scoremap          PET/dataset/pose_defaultdataset.py:220-266
  coord2map                             DGP/dataset.py:246-271
  compute_pred_dims                     DGP/dataset.py:348-371 (closed form, no network run)

Marker id convention: marker = frame_in_batch * nj + joint.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from .arch import scoremap_hw

_EMPTY = np.empty(0, dtype="int")


def _nan_markers(frames: np.ndarray, joint_loc: np.ndarray) -> np.ndarray:
    """Marker ids (sorted) of NaN-labelled joints inside the visible frames."""
    frames = np.asarray(frames)
    if joint_loc.shape[0] == 0:
        return _EMPTY
    nj = joint_loc.shape[1]
    fi, ji = np.nonzero(np.isnan(joint_loc[:, :, 0]))
    return np.sort(frames[fi] * nj + ji).astype("int")


def _all_markers(frames: np.ndarray, nj: int) -> np.ndarray:
    frames = np.asarray(frames).astype("int")
    return np.sort((frames[:, None] * nj + np.arange(nj)[None, :]).ravel())


def gen_idx_chunk(visible_frame_indices, hidden_frame_indices, joint_loc):
    """-> (visible_marker, hidden_marker, visible_marker_in_targets); DGP/dataset.py:187-239.

    NaN-labelled joints of visible frames move to the hidden set; visible_marker_in_targets indexes
    the flattened [n_visible_frames * nj] label array."""
    visible_frame_indices = np.asarray(visible_frame_indices)
    hidden_frame_indices = np.asarray(hidden_frame_indices)
    nj = joint_loc.shape[1]
    nan_ind = _nan_markers(visible_frame_indices, joint_loc)
    hidden = np.sort(np.concatenate([_all_markers(hidden_frame_indices, nj), nan_ind])).astype("int")
    vis0 = _all_markers(visible_frame_indices, nj)
    keep = ~np.isin(vis0, nan_ind)
    visible = vis0[keep]
    if visible.size == 0:
        return _EMPTY, (hidden if hidden.size else _EMPTY), _EMPTY
    return visible, (hidden if hidden.size else _EMPTY), np.nonzero(keep)[0]


def find_marker_index(pv, ph, joint_loc):
    """-> (pv_ts, ph_ts): visible / hidden marker ids of a chunk; DGP/dataset.py:157-184."""
    nj = joint_loc.shape[1]
    nan_ind = _nan_markers(np.asarray(pv), joint_loc)
    ph_ts = np.sort(np.concatenate([_all_markers(np.asarray(ph), nj), nan_ind]))
    pv_ts = np.setdiff1d(_all_markers(np.asarray(pv), nj), nan_ind)
    return pv_ts, ph_ts


def make_neighboring_window(window_size: int = 5) -> np.ndarray:
    return np.arange(-window_size, window_size + 1)


def get_neighboring_window(pv_all, ns: int, nt_max: int, nt_min: int = 0) -> np.ndarray:
    """Union of [p-ns, p+ns] over p in pv_all, clipped to [nt_min, nt_max); DGP/dataset.py:113-119."""
    pv_all = np.asarray(pv_all)
    w = np.unique(pv_all[:, None] + make_neighboring_window(ns)[None, :])
    return w[(w >= nt_min) & (w < nt_max)]


def select_hidden_frames(ns, pv_all, pvh_sorted, n_frames, n_max_frames, ns_jump=None, verbose: bool = False):
    """Greedy pick of high-motion-energy unlabeled frames; DGP/dataset.py:46-101.

    pvh_sorted: frame ids sorted by decreasing motion energy.  A candidate is skipped if it lies in the
    +-ns window of a labeled frame or within ns_small = max(ns - ns_jump, 1) of an already chosen frame;
    selection stops when the union of windows would exceed n_max_frames."""
    if ns_jump is None:
        ns_jump = ns
    ns_small = max(ns - ns_jump, 1)
    pv_all = np.asarray(pv_all)
    pv_windowed = get_neighboring_window(pv_all, ns, n_frames)
    ph_all = np.empty(0, dtype="int")
    if len(pv_windowed) >= n_max_frames:
        if verbose:
            print("Visible frames + window exceed n_max_frames; skipping selection of hidden frames")
        return ph_all
    candidates = np.asarray(pvh_sorted)[~np.isin(pvh_sorted, pv_windowed)]
    chosen = pv_all.copy()
    n_sel = n_skip = 0
    for cand in candidates:
        if chosen.size and np.abs(cand - chosen).min() < ns_small:
            n_skip += 1
            continue
        if len(get_neighboring_window(np.append(chosen, cand), ns, n_frames)) > n_max_frames:
            break
        ph_all = np.append(ph_all, cand)
        chosen = np.append(chosen, cand)
        n_sel += 1
    if verbose:
        print("Selected additional {} hidden frames".format(n_sel))
        print("Skipped {} high motion energy (me) frames since in visible window or close to higher me "
              "hidden frame".format(n_skip))
    return ph_all


def compute_target_part_scoremap(joint_id, coords, size, num_joints: int, pos_dist_thresh: float,
                                 stride: float = 8.0, locref_stdev: float = 7.2801, scale: float = 1.0):
    """DLC target maps for one image; PET/dataset/pose_defaultdataset.py:220-266.

    joint_id: list (per animal) of joint ids; coords: list of [k,2] (x, y) pixel coordinates.
    -> (scmap [H,W,nj], locref_map [H,W,2nj], locref_mask [H,W,2nj]); a cell belongs to a joint when its
    centre (i*stride + stride/2) is within pos_dist_thresh*scale px; locref = (dx, dy) / locref_stdev."""
    h, w = int(size[0]), int(size[1])
    half = stride / 2.0
    thr = pos_dist_thresh * scale
    thr_sq = thr ** 2
    scmap = np.zeros((h, w, num_joints))
    lmap = np.zeros((h, w, 2 * num_joints))
    lmask = np.zeros((h, w, 2 * num_joints))
    for person in range(len(coords)):
        for k, j_id in enumerate(joint_id[person]):
            j_x, j_y = float(coords[person][k, 0]), float(coords[person][k, 1])
            cx = round((j_x - half) / stride)
            cy = round((j_y - half) / stride)
            x0, x1 = round(max(cx - thr - 1, 0)), round(min(cx + thr + 1, w - 1))
            y0, y1 = round(max(cy - thr - 1, 0)), round(min(cy + thr + 1, h - 1))
            if x1 < x0 or y1 < y0:
                continue
            dx = j_x - (np.arange(x0, x1 + 1) * stride + half)
            dy = j_y - (np.arange(y0, y1 + 1) * stride + half)
            inside = (dx[None, :] ** 2 + dy[:, None] ** 2) <= thr_sq
            sl = (slice(y0, y1 + 1), slice(x0, x1 + 1))
            scmap[sl + (j_id,)][inside] = 1
            for ch, val in ((2 * j_id, np.broadcast_to(dx[None, :], inside.shape)),
                            (2 * j_id + 1, np.broadcast_to(dy[:, None], inside.shape))):
                lmask[sl + (ch,)][inside] = 1
                lmap[sl + (ch,)][inside] = val[inside] * (1.0 / locref_stdev)
    return scmap, lmap, lmask


def coord2map(joint_loc, nx_out: int, ny_out: int, nj: int, pos_dist_thresh: float,
              locref_stdev: float = 7.2801):
    """Locref targets / masks for the labeled frames of a batch; DGP/dataset.py:246-271.

    joint_loc [nv, nj, 2] in scoremap units (row, col), NaN = unlabeled.  The reference hard-codes
    *8 + 4 for the pixel conversion and calls the DLC target generator with scale = 1; joints whose
    (NaN->0) coordinates sum to zero are dropped."""
    targets, masks = [], []
    for ii in range(joint_loc.shape[0]):
        xy = np.flip(joint_loc[ii] * 8 + 4, 1)            # (row, col) -> (x, y) px
        present = np.where(np.nan_to_num(xy).sum(1) != 0)[0]
        _, lt, lm = compute_target_part_scoremap([present], [xy[present]], (nx_out, ny_out), nj, pos_dist_thresh,
                                                 8.0, locref_stdev, 1.0)
        targets.append(lt)
        masks.append(lm)
    t = np.array(targets).squeeze()
    m = np.array(masks).squeeze()
    if t.ndim == 3:
        t, m = t[None], m[None]
    return t, m


def compute_pred_dims(frame_h: int, frame_w: int) -> Tuple[int, int]:
    """(nx_out, ny_out) of the scoremap for a frame: 2*ceil(H/16), 2*ceil(W/16).  The reference builds and
    runs the whole network on a zero frame just to read this shape (DGP/dataset.py:348-371)."""
    return scoremap_hw(frame_h, frame_w)
