"""Host-side training-sample loader of DLC's step-0 trainer.

Counterpart of `PoseDataset` in DeepLabCut pose_estimation_tensorflow/dataset/pose_defaultdataset.py:19-276 and
`CropImage` in dataset/pose_dataset.py:37-52, which fit_dlc (DGP/models/fitdgp.py:119) feeds through a TF queue.
Same random-number call order (python `random.uniform` for the scale, `np.random` for the shuffle, the crop
decision, the crop joint and the four crop extents), so equal seeds walk the same sample schedule.

Differences, both forced by the image: cv2 is absent, so `imresize` averages with PIL's BOX filter instead of
cv2.INTER_AREA (same support, slightly different rounding of the output size: both use round-half-even of
size*scale here); mirroring needs `all_joints` symmetric pairs and is carried over unchanged.
"""
from __future__ import annotations

import os
import random as rand

import numpy as np

from .dataset import compute_target_part_scoremap


class DataItem:
    pass


def imread(path: str) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


def imresize(img: np.ndarray, scale: float) -> np.ndarray:
    """Area-averaging resize by a factor (auxfun_videos.imresize, utils/auxfun_videos.py:21-25)."""
    if scale == 1.0:
        return img
    try:
        import cv2
        return cv2.resize(img, None, fx=scale, fy=scale, interpolation=cv2.INTER_AREA)
    except ImportError:
        from PIL import Image
        h, w = img.shape[:2]
        nw, nh = max(1, int(round(w * scale))), max(1, int(round(h * scale)))
        return np.asarray(Image.fromarray(img).resize((nw, nh), Image.BOX if scale < 1 else Image.BILINEAR))


def mirror_joints_map(all_joints, num_joints):
    res = np.arange(num_joints)
    symmetric = [p for p in all_joints if len(p) == 2]
    for a, b in symmetric:
        res[a], res[b] = b, a
    return res


def crop_image(joints, im, x_label, y_label, cfg):
    """Random crop around one labeled joint (pose_dataset.py:37-52)."""
    fwd = int(cfg["minsize"] + np.random.randint(cfg["rightwidth"]))
    back = int(cfg["minsize"] + np.random.randint(cfg["leftwidth"]))
    hup = int(cfg["minsize"] + np.random.randint(cfg["topheight"]))
    hdown = int(cfg["minsize"] + np.random.randint(cfg["bottomheight"]))
    x0 = max(0, int(x_label - back))
    x1 = min(im.shape[1] - 1, int(x_label + fwd))
    y0 = max(0, int(y_label - hdown))
    y1 = min(im.shape[0] - 1, int(y_label + hup))
    joints[0, :, 1] -= x0
    joints[0, :, 2] -= y0
    # like the reference, the bounds test uses the un-cropped image size
    inb = np.where((joints[0, :, 1] > 0) * (joints[0, :, 1] < im.shape[1]) * (joints[0, :, 2] > 0) *
                   (joints[0, :, 2] < im.shape[0]))[0]
    return joints[:, inb, :], im[y0:y1 + 1, x0:x1 + 1, :]


class PoseDataset:
    def __init__(self, cfg):
        self.cfg = cfg
        self.data = self.load_dataset()
        self.num_images = len(self.data)
        self.max_input_sizesquare = cfg.get("max_input_size", 1500) ** 2
        self.min_input_sizesquare = cfg.get("min_input_size", 64) ** 2
        self.stride = cfg.stride
        self.scale = cfg.global_scale
        self.scale_jitter_lo = cfg.get("scale_jitter_lo", .75)
        self.scale_jitter_up = cfg.get("scale_jitter_up", 1.25)
        if cfg.mirror:
            self.symmetric_joints = mirror_joints_map(cfg.all_joints, cfg.num_joints)
        self.curr_img = 0
        self.shuffle = cfg.shuffle
        if not self.shuffle:
            assert not cfg.mirror
            self.image_indices = np.arange(self.num_images)

    def load_dataset(self):
        import scipy.io as sio
        cfg = self.cfg
        mlab = sio.loadmat(os.path.join(cfg.project_path, cfg.dataset))["dataset"]
        data = []
        self.has_gt = True
        for i in range(mlab.shape[1]):
            sample = mlab[0, i]
            item = DataItem()
            item.image_id = i
            item.im_path = str(sample[0][0])
            item.im_size = sample[1][0]
            if len(sample) >= 3:
                joints = np.asarray(sample[2][0][0])
                if joints.size:
                    assert (joints[:, 0] < cfg.num_joints).any()
                item.joints = [joints]
            else:
                self.has_gt = False
            data.append(item)
        return data

    def num_training_samples(self):
        return self.num_images * (2 if self.cfg.mirror else 1)

    def shuffle_images(self):
        n = self.num_images
        if self.cfg.mirror:
            idx = np.random.permutation(n * 2)
            self.mirrored = idx >= n
            idx[self.mirrored] = idx[self.mirrored] - n
            self.image_indices = idx
        else:
            self.image_indices = np.random.permutation(n)

    def next_training_sample(self):
        if self.curr_img == 0 and self.shuffle:
            self.shuffle_images()
        cur = self.curr_img
        self.curr_img = (self.curr_img + 1) % self.num_training_samples()
        return self.image_indices[cur], bool(self.cfg.mirror and self.mirrored[cur])

    def get_scale(self):
        return rand.uniform(self.scale_jitter_lo, self.scale_jitter_up) * self.scale

    def is_valid_size(self, image_size, scale):
        area = (image_size[2] * scale) * (image_size[1] * scale)
        return self.min_input_sizesquare <= area <= self.max_input_sizesquare

    def next_batch(self):
        while True:
            imidx, mirror = self.next_training_sample()
            item = self.data[imidx]
            scale = self.get_scale()
            if self.is_valid_size(item.im_size, scale):
                return self.make_batch(item, scale, mirror)

    def skip_batch(self):
        """Advance every random stream exactly as next_batch() would -- sample index, mirror flag, scale jitter with its
        size-rejection loop, the crop decision and its five draws -- WITHOUT reading, scaling or cropping the image and without
        building target maps.  Data-parallel fit_dlc: rank r trains on sample it * W + r of the common sequence and skips the
        other W - 1 samples of an iteration, so its host work per step stays that of one sample."""
        cfg = self.cfg
        while True:
            imidx, mirror = self.next_training_sample()
            item = self.data[imidx]
            scale = self.get_scale()
            if self.is_valid_size(item.im_size, scale):
                break
        if self.has_gt and cfg.crop and np.random.rand() < cfg.cropratio and np.asarray(item.joints).shape[1] > 0:
            np.random.randint(np.asarray(item.joints).shape[1])
            for key in ("rightwidth", "leftwidth", "topheight", "bottomheight"):      # crop_image's draws, in its order
                np.random.randint(cfg[key])

    def make_batch(self, item, scale, mirror):
        """-> dict(inputs uint8 [1,H,W,3], part_score_targets / part_score_weights [1,h,w,nj],
        locref_targets / locref_mask [1,h,w,2nj], data_item)."""
        cfg = self.cfg
        image = imread(os.path.join(cfg.project_path, item.im_path))
        joints = np.array(item.joints, dtype=np.float64) if self.has_gt else None      # [1, k, 3] copy
        if self.has_gt and cfg.crop and np.random.rand() < cfg.cropratio and joints.shape[1] > 0:
            j = np.random.randint(joints.shape[1])
            joints, image = crop_image(joints, image, joints[0, j, 1], joints[0, j, 2], cfg)
        img = imresize(image, scale) if scale != 1 else image
        size = np.array(img.shape[0:2])
        if mirror:
            img = np.fliplr(img)
        batch = {"inputs": np.array(img)[None]}
        if self.has_gt:
            if mirror:
                jm = []
                for pj in joints:
                    r = np.copy(pj)
                    r[:, 1] = image.shape[1] - r[:, 1] - 1
                    r[:, 0] = self.symmetric_joints[pj[:, 0].astype(int)]
                    jm.append(r)
                joints = jm
            sm_size = np.ceil(size / (self.stride * 2)).astype(int) * 2
            coords = [pj[:, 1:3] * scale for pj in joints]
            ids = [pj[:, 0].astype(int) for pj in joints]
            scmap, lmap, lmask = compute_target_part_scoremap(ids, coords, sm_size, cfg.num_joints, cfg.pos_dist_thresh,
                                                              stride=self.stride, locref_stdev=cfg.locref_stdev,
                                                              scale=scale)
            if cfg.weigh_only_present_joints:
                weights = np.zeros(scmap.shape)
                for pid in ids:
                    weights[:, :, pid] = 1.0
            else:
                weights = np.ones(scmap.shape)
            batch.update(part_score_targets=scmap[None].astype(np.float32), part_score_weights=weights[None].astype(np.float32),
                         locref_targets=lmap[None].astype(np.float32), locref_mask=lmask[None].astype(np.float32))
        batch["data_item"] = item
        return batch


class LearningRate:
    """Multi-step schedule (train.py:34-44): lr of the current segment; the segment advances when the
    iteration equals its end."""

    def __init__(self, cfg):
        self.steps = cfg.multi_step
        self.current_step = 0

    def get_lr(self, iteration):
        lr = self.steps[self.current_step][0]
        if iteration == self.steps[self.current_step][1]:
            self.current_step += 1
        return lr
