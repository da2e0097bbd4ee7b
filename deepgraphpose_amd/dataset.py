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


# =====================================================================================================
# Dataset / MultiDataset: per-video bookkeeping of the training batches (DGP/dataset.py:305-1036)
# =====================================================================================================
import copy
import os
from os.path import join, split
from pathlib import Path


def calculate_motion_energy(source, backend: str = "auto", chunk: int = 256) -> np.ndarray:
    """mean |frame_t - frame_{t-1}| per frame (DGP/dataset.py:29-43).  Frames are uint8 and the reference
    subtracts them as such, so the difference wraps modulo 256 -- restated as is.

    backend "hip": chunks of `chunk` frames are staged to the GPU and reduced by `dgp_motion_energy` (exact integer sums, so the
    result is bit-identical to the numpy loop); "host": the reference's per-frame numpy loop; "auto": "hip" when a GPU is
    visible (and then the HIP library must load -- no silent fallback), "host" on boxes without one."""
    if backend == "auto":
        import torch
        backend = "hip" if torch.cuda.is_available() else "host"
    if backend == "host":
        me, prev, n = [], None, 0
        for frame in source.iter_frames():
            frame = np.asarray(frame)
            me.append(0.0 if prev is None else float(np.mean(np.abs(frame - prev))))
            prev = frame
            n += 1
        return np.asarray(me[:n])
    if backend != "hip":
        raise ValueError("calculate_motion_energy: backend must be auto | hip | host, not %r" % (backend,))
    import torch
    from . import engine
    out, prev, pinned = [], None, None

    def batches():
        if hasattr(source, "iter_batches"):
            yield from source.iter_batches(chunk)
            return
        buf = []
        for frame in source.iter_frames():
            buf.append(np.asarray(frame))
            if len(buf) == chunk:
                yield np.stack(buf); buf = []
        if buf:
            yield np.stack(buf)

    for b in batches():
        b = np.ascontiguousarray(b, dtype=np.uint8)
        if pinned is None or pinned.shape[0] < b.shape[0] or tuple(pinned.shape[1:]) != tuple(b.shape[1:]):
            pinned = torch.empty((max(chunk, b.shape[0]),) + tuple(b.shape[1:]), dtype=torch.uint8).pin_memory()
        np.copyto(pinned.numpy()[:b.shape[0]], b)
        dev = pinned[:b.shape[0]].to("cuda", non_blocking=True)
        out.append(engine.motion_energy(dev, prev))
        prev = dev[-1].clone()
    return np.concatenate(out) if out else np.zeros(0)


def get_frame_idxs_from_train_mat(data_array, video: str) -> np.ndarray:
    """Frame numbers (img<NNN>.png) of the training items that belong to `video` (DGP/dataset.py:272-282)."""
    idxs = []
    for dat in data_array:
        path = str(dat[0][0])
        if video in os.path.normpath(path).split(os.sep):
            idxs.append(int(split(path)[-1][3:].split(".")[0]))
    return np.sort(idxs)


def load_train_mat_labels(mat_file: str, video: str, nj: int, stride: float = 8.0):
    """Labels of one video from DLC's training .mat (PET/dataset/pose_defaultdataset.py:39-76) converted like
    Dataset._compute_targets (DGP/dataset.py:587-657): targets_2d = flip((xy - stride/2) / stride), NaN = unlabeled.
    -> (targets_2d [nv, nj, 2] (row, col) in scoremap units, frame_idxs [nv])."""
    import scipy.io as sio
    data = sio.loadmat(mat_file)["dataset"][0]
    frames, targets = [], []
    for dat in data:
        path = str(dat[0][0])
        if video not in os.path.normpath(path).split(os.sep):
            continue
        idx = int(split(path)[-1][3:].split(".")[0])
        if idx in frames:
            continue
        t = np.full((nj, 2), np.nan)
        if len(dat) >= 3:
            joints = dat[2][0][0]
            for row in joints:
                t[int(row[0])] = np.flip((row[1:3].astype(np.float64) - stride / 2) / stride)
        frames.append(idx)
        targets.append(t)
    order = np.argsort(frames)
    if not frames:
        return np.empty((0, nj, 2)), np.array([], dtype=int)
    return np.asarray(targets)[order], np.asarray(frames)[order]


class Dataset:
    """One video: frame source, labeled-frame indices, selected hidden frames, label targets."""

    def __init__(self, video_path, dlc_config, paths, source=None, labels=None, label_idxs=None):
        from .frames import open_frame_source
        self.video_path = video_path
        self.video_name = os.path.basename(str(video_path)).rpartition(".")[0] or os.path.basename(str(video_path))
        self.video_clip = source if source is not None else open_frame_source(video_path)
        self.video_n_frames = self.n_frames = int(self.video_clip.n_frames)
        self.dlc_config = dlc_config
        self.paths = copy.deepcopy(paths)
        self.nj = dlc_config.num_joints
        self.ny_in, self.nx_in = self.video_clip.size           # (width, height) -> nx_in = rows
        self.nx_out, self.ny_out = compute_pred_dims(self.nx_in, self.ny_in)
        if labels is None:
            mat = join(dlc_config["project_path"], dlc_config["dataset"])
            labels, label_idxs = load_train_mat_labels(mat, self.video_name, self.nj, dlc_config.get("stride", 8.0))
        self.labels_all, self.labels_idxs_all = np.asarray(labels), np.asarray(label_idxs, dtype=int)
        self.idxs = {"vis": {"train": np.sort(self.labels_idxs_all), "val": np.array([])}}
        self.curr_batch = 0
        self.batch_data = None
        self.global_offset = 0
        self._cache = {}

    # -- batch creation ------------------------------------------------------------------------------
    def create_batches_from_resnet_output(self, batch_info, batches_path):
        """Name kept from the reference (DGP/dataset.py:373-424); nothing is pushed through a ResNet here --
        it selects the hidden frames and builds the index sets."""
        self.paths["batched_data"] = Path(batches_path)
        self.batch_key = "nsjump=%s_step=%i_ns=%i_nc=%i_max=%i" % (batch_info["ns_jump"], batch_info["step"],
                                                                   batch_info["ns"], batch_info["nc"],
                                                                   batch_info["n_max_frames"])
        pv_idxs = self.idxs["vis"]["train"]
        ph_idxs = self._find_good_hidden_frames(pv_idxs, batch_info)
        self.idxs["pv"], self.idxs["ph"] = np.array(pv_idxs), np.array(ph_idxs, dtype=int)
        chunk_id = np.concatenate([self.idxs["pv"], self.idxs["ph"]]).astype(int)
        ns_new = np.ceil(batch_info["n_max_frames"] / max(len(chunk_id), 1) / 2)
        ns_new = int(min(ns_new, batch_info["ns"]))
        self.idxs["chunk"] = get_neighboring_window(chunk_id, ns_new, self.video_n_frames)
        self.idxs["pv_chunk"] = np.where(np.isin(self.idxs["chunk"], pv_idxs))[0]
        self.idxs["ph_chunk"] = np.where(np.isin(self.idxs["chunk"], ph_idxs))[0]
        self.idxs["ph_all_chunk"] = np.where(~np.isin(self.idxs["chunk"], pv_idxs))[0]
        self.labels, self.labels_idxs = self.labels_all, self.labels_idxs_all

    def _find_good_hidden_frames(self, pv_idxs, batch_info):
        """High-motion-energy unlabeled frames, cached as .npy next to the model (DGP/dataset.py:517-556)."""
        idxs_file = Path(self.paths["batched_data"]) / ("%s__%s_idxs.npy" % (self.video_name, self.batch_key))
        if os.path.exists(idxs_file):
            idxs = np.load(idxs_file, allow_pickle=True).item()
            if len(idxs["pv"]) == len(pv_idxs) and np.all(np.sort(pv_idxs) == np.sort(idxs["pv"])):
                return idxs["ph"]
        me = calculate_motion_energy(self.video_clip)
        order = np.argsort(me).flatten()[::-1]
        sel = np.sort(select_hidden_frames(batch_info["ns"], np.asarray(pv_idxs), order, self.video_n_frames,
                                           batch_info["n_max_frames"], batch_info["ns_jump"]))
        sel = sel[np.arange(0, len(sel), batch_info["step"]).astype(int)]
        os.makedirs(os.path.dirname(idxs_file), exist_ok=True)
        np.save(idxs_file, {"pv": np.asarray(pv_idxs), "ph": sel})
        return sel

    def reset(self):
        np.random.shuffle(self.idxs["pv"])
        np.random.shuffle(self.idxs["ph"])
        self.curr_batch = 0

    # -- batches -------------------------------------------------------------------------------------
    def load_data(self, idxs_video, pv_idxs):
        """Random-access frames + the labels of the visible ones (DGP/dataset.py:811-821)."""
        got = {i: np.asarray(self.video_clip.frame_at(i)) for i in set(int(i) for i in idxs_video)}
        images = np.stack([got[int(i)] for i in idxs_video]).astype(np.uint8)
        idxs_labels = [int(np.where(self.labels_idxs == i)[0][0]) for i in pv_idxs]
        return images, self.labels[idxs_labels]

    def next_batch(self, schedule, batch_dict, pv_idxs=None, ph_idxs=None):
        """-> (pv_idxs, ph_idxs, pv_idxs_b, images u8 [nt,H,W,3], labels [nv,nj,2], batch_mask [nt-1], batch_ts,
        (visible_marker, hidden_marker, visible_marker_in_targets)); DGP/dataset.py:672-750."""
        if pv_idxs is None and ph_idxs is None:
            raise NotImplementedError("schedule-driven batching is legacy; fit_dgp passes explicit frame lists")
        pv_idxs, ph_idxs = np.asarray(pv_idxs, dtype=int), np.asarray(ph_idxs, dtype=int)
        idxs_video = np.sort(np.concatenate([pv_idxs, ph_idxs]))
        images, labels = self.load_data(idxs_video, pv_idxs)
        pv_idxs_b = np.where(np.isin(idxs_video, pv_idxs))[0]
        ph_idxs_b = np.where(np.isin(idxs_video, ph_idxs))[0]
        batch_mask = np.zeros(max(len(idxs_video) - 1, 0), dtype=int)
        batch_mask[np.where(np.diff(idxs_video) == 1)[0]] = 1
        pv_chunk = np.where(np.isin(self.idxs["chunk"], pv_idxs))[0]
        ph_chunk = np.where(np.isin(self.idxs["chunk"], ph_idxs))[0]
        pv_ts, ph_ts = find_marker_index(pv_chunk, ph_chunk, labels)
        batch_ts = self.global_offset * self.nj + np.unique(list(pv_ts) + list(ph_ts))
        addn = gen_idx_chunk(pv_idxs_b, ph_idxs_b, labels)
        self.batch_data = (pv_idxs, ph_idxs, pv_idxs_b, images, labels, batch_mask, batch_ts, addn)
        return self.batch_data


class MultiDataset:
    """All videos of a project (DGP/dataset.py:824-1036)."""

    def __init__(self, config_yaml, video_sets=None, shuffle=None, S0=None, sources=None):
        import yaml
        from .config import get_train_config
        self.datasets, self.paths = [], {}
        with open(config_yaml, "r") as stream:
            self.proj_config = yaml.safe_load(stream)
        if video_sets is None:
            self.proj_config["video_path"] = self.proj_config["video_sets"]
        else:
            keys = [split(v)[-1] for v in self.proj_config["video_sets"].keys()]
            if set(keys) == set(split(v)[-1] for v in video_sets):
                self.proj_config["video_path"] = self.proj_config["video_sets"]
            else:
                self.proj_config["video_path"] = {v: {} for v in video_sets}
                self.proj_config["video_sets"] = {v: {} for v in video_sets}
        self.proj_config["video_sets"] = {join(self.proj_config["project_path"], k): v
                                          for k, v in self.proj_config["video_sets"].items()}
        self.dlc_config = get_train_config(self.proj_config, shuffle)
        self.paths["project"] = Path(self.dlc_config.project_path)
        self.paths["dlc_model"] = Path(self.dlc_config.snapshot_prefix).parent
        self.paths["batched_data"] = ""
        self.video_files = list(self.proj_config["video_sets"].keys())
        assert len(self.video_files) > 0
        ratios = []
        for i, vf in enumerate(self.video_files):
            src = None if sources is None else sources[i]
            self.datasets.append(Dataset(vf, self.dlc_config, self.paths, source=src))
            ratios.append(len(self.datasets[-1].idxs["vis"]["train"]))
        self.batch_ratios = np.array(ratios) / max(np.sum(ratios), 1)
        self.n_datasets = len(self.datasets)
        self.nj = self.datasets[0].nj
        self.S0 = S0
        self.nx_in = self.ny_in = self.nx_out = self.ny_out = None
        self.n_visible_frames_total = self.n_hidden_frames_total = self.n_frames_total = 0
        self.curr_batch = 0

    def __len__(self):
        return self.n_datasets

    def create_batches_from_resnet_output(self, snapshot, ns_jump=None, ns=10, nc=200, step=2, n_max_frames=1000):
        self.snapshot = snapshot
        self.batch_info = dict(ns_jump=ns_jump, ns=ns, nc=nc, step=step, n_max_frames=n_max_frames)
        self.paths["batched_data"] = self.paths["dlc_model"] / "batched_data" / "snapshot-{}".format(snapshot)
        for d in self.datasets:
            d.create_batches_from_resnet_output(self.batch_info, self.paths["batched_data"])
        d0 = self.datasets[0]
        self.nx_in, self.ny_in, self.nx_out, self.ny_out = d0.nx_in, d0.ny_in, d0.nx_out, d0.ny_out
        self.n_visible_frames_total = self.n_hidden_frames_total = self.n_frames_total = 0
        for d in self.datasets:
            self.n_visible_frames_total += len(d.idxs["pv"])
            self.n_hidden_frames_total += len(d.idxs["ph"])
            d.global_offset = self.n_frames_total
            self.n_frames_total += len(d.idxs["chunk"])

    def reset(self):
        for d in self.datasets:
            d.reset()
        self.curr_batch = 0

    def next_batch(self, schedule, dataset=None, pv_idxs=None, ph_idxs=None):
        if dataset is None or pv_idxs is None or ph_idxs is None:
            raise NotImplementedError("schedule-driven batching is legacy; pass dataset, pv_idxs, ph_idxs")
        return self.datasets[dataset].next_batch(schedule, self.batch_info, pv_idxs=pv_idxs, ph_idxs=ph_idxs), dataset
