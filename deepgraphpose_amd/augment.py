"""Host-side augmentation of the labeled frames of a DGP batch.

Counterpart of `build_aug` / `data_aug` (DGP/models/fitdgp_util.py:412-451): the reference chains seven imgaug
augmenters, each behind `Sometimes(apply_prob)`, over the VISIBLE frames of a batch and moves the labels with the
pixels.  imgaug is a third-party host library the reference only calls; when it is importable `build_aug` returns the
reference's own pipeline object, otherwise the numpy / scipy pipeline below, which draws the same parameters from the
same distributions (own RNG stream: imgaug's bit stream is not reproducible without imgaug):

  Fliplr(0.5) | Affine(rotate=(-10, 10)) | MotionBlur(k=3, angle=(-90, 90)) | CoarseDropout((0, 0.02), size_percent=(0.01, 0.05))
  | ElasticTransformation(sigma=5, alpha=(0, 10)) | AdditiveGaussianNoise(scale=(0, 0.01*255), per_channel=0.5)
  | Sometimes(0.4, CropAndPad(percent=(-0.3, 0.1), keep_size=True))

Keypoints are (x, y) pixel coordinates, NaN = unlabeled (NaNs stay NaN).  Both pipelines are called as
`pipeline(images=uint8 [n,H,W,3], keypoints=[[(x, y), ...], ...]) -> (images, keypoints)`.
"""
from __future__ import annotations

import numpy as np


class NumpyAugPipeline:
    def __init__(self, apply_prob: float = 0.5, seed=None):
        self.p = float(apply_prob)
        self.rng = np.random.RandomState(seed)

    # ---- the seven augmenters: (image uint8 [H,W,C], keypoints float [nj,2] as (x, y)) -> the same ----
    def fliplr(self, img, kp):
        if self.rng.random_sample() < 0.5:
            img = img[:, ::-1]
            kp = kp.copy()
            kp[:, 0] = img.shape[1] - kp[:, 0]
        return img, kp

    def rotate(self, img, kp):
        from scipy import ndimage
        deg = self.rng.uniform(-10, 10)
        t = np.deg2rad(deg)
        H, W = img.shape[:2]
        c = np.array([(W - 1) / 2.0, (H - 1) / 2.0])                      # rotation about the image centre, (x, y)
        R = np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]])   # output = R (input - c) + c in (x, y)
        Ri = R.T                                                          # inverse map for the resampler
        M = Ri[::-1, ::-1]                                                # (row, col) ordering
        off = c[::-1] - M @ c[::-1]
        out = np.stack([ndimage.affine_transform(img[..., ch].astype(np.float32), M, offset=off, order=1, mode="constant",
                                                 cval=0.0) for ch in range(img.shape[2])], -1)
        kp2 = (kp - c) @ R.T + c
        return np.clip(np.rint(out), 0, 255).astype(np.uint8), kp2

    def motion_blur(self, img, kp):
        from scipy import ndimage
        ang = np.deg2rad(self.rng.uniform(-90, 90))
        k = np.zeros((3, 3), dtype=np.float32)
        k[1, 1] = 1.0
        dx, dy = np.cos(ang), np.sin(ang)
        for s in (-1, 1):                                                 # a 3-tap line through the centre at that angle (bilinear weights)
            x, y = 1 + s * dx, 1 + s * dy
            x0, y0 = int(np.floor(x)), int(np.floor(y))
            for yy, wy in ((y0, 1 - (y - y0)), (y0 + 1, y - y0)):
                for xx, wx in ((x0, 1 - (x - x0)), (x0 + 1, x - x0)):
                    if 0 <= yy < 3 and 0 <= xx < 3:
                        k[yy, xx] += wy * wx
        k /= k.sum()
        out = np.stack([ndimage.convolve(img[..., ch].astype(np.float32), k, mode="reflect") for ch in range(img.shape[2])], -1)
        return np.clip(np.rint(out), 0, 255).astype(np.uint8), kp

    def coarse_dropout(self, img, kp):
        H, W = img.shape[:2]
        p = self.rng.uniform(0, 0.02)
        sp = self.rng.uniform(0.01, 0.05)
        h, w = max(1, int(round(H * sp))), max(1, int(round(W * sp)))
        drop = self.rng.random_sample((h, w)) < p
        rows = np.minimum((np.arange(H) * h) // H, h - 1)
        cols = np.minimum((np.arange(W) * w) // W, w - 1)
        mask = drop[rows][:, cols]                                        # nearest-neighbour upsampling of the coarse mask
        out = img.copy()
        out[mask] = 0
        return out, kp

    def elastic(self, img, kp):
        from scipy import ndimage
        H, W = img.shape[:2]
        alpha = self.rng.uniform(0, 10)
        dx = ndimage.gaussian_filter(self.rng.random_sample((H, W)) * 2 - 1, 5, mode="constant") * alpha
        dy = ndimage.gaussian_filter(self.rng.random_sample((H, W)) * 2 - 1, 5, mode="constant") * alpha
        yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        coords = np.array([yy + dy, xx + dx])
        out = np.stack([ndimage.map_coordinates(img[..., ch].astype(np.float32), coords, order=1, mode="constant", cval=0.0)
                        for ch in range(img.shape[2])], -1)
        kp2 = kp.copy()
        for j in range(kp.shape[0]):                                      # out[p] = in[p + d(p)]  =>  a point moves by about -d
            x, y = kp[j]
            if np.isfinite(x) and np.isfinite(y):
                xi, yi = int(np.clip(round(x), 0, W - 1)), int(np.clip(round(y), 0, H - 1))
                kp2[j, 0] = x - dx[yi, xi]
                kp2[j, 1] = y - dy[yi, xi]
        return np.clip(np.rint(out), 0, 255).astype(np.uint8), kp2

    def gaussian_noise(self, img, kp):
        scale = self.rng.uniform(0.0, 0.01 * 255)
        per_channel = self.rng.random_sample() < 0.5
        shape = img.shape if per_channel else img.shape[:2] + (1,)
        out = img.astype(np.float32) + self.rng.normal(0.0, scale, size=shape)
        return np.clip(np.rint(out), 0, 255).astype(np.uint8), kp

    def crop_and_pad(self, img, kp):
        from PIL import Image
        H, W = img.shape[:2]
        top, right, bottom, left = self.rng.uniform(-0.3, 0.1, size=4)    # one draw per side; < 0 crops, > 0 pads (zeros)
        t, b = int(round(top * H)), int(round(bottom * H))
        l, r = int(round(left * W)), int(round(right * W))
        # never crop the image away (imgaug keeps at least one pixel per axis)
        if -(t + b) >= H:
            t = b = -((H - 1) // 2)
        if -(l + r) >= W:
            l = r = -((W - 1) // 2)
        x0, x1 = max(0, -l), W - max(0, -r)
        y0, y1 = max(0, -t), H - max(0, -b)
        core = img[y0:y1, x0:x1]
        pt, pb, pl_, pr = max(0, t), max(0, b), max(0, l), max(0, r)
        core = np.pad(core, ((pt, pb), (pl_, pr), (0, 0)), mode="constant")
        h2, w2 = core.shape[:2]
        out = np.asarray(Image.fromarray(core).resize((W, H), Image.BILINEAR))    # keep_size=True
        kp2 = kp.copy()
        kp2[:, 0] = (kp[:, 0] - x0 + pl_) * (W / float(w2))
        kp2[:, 1] = (kp[:, 1] - y0 + pt) * (H / float(h2))
        return out, kp2

    def __call__(self, images, keypoints):
        out_imgs, out_kps = [], []
        chain = [(self.p, self.fliplr), (self.p, self.rotate), (self.p, self.motion_blur), (self.p, self.coarse_dropout),
                 (self.p, self.elastic), (self.p, self.gaussian_noise), (0.4, self.crop_and_pad)]
        for img, kps in zip(images, keypoints):
            img = np.ascontiguousarray(img, dtype=np.uint8)
            kp = np.asarray(kps, dtype=np.float64).reshape(-1, 2)
            for prob, fn in chain:
                if self.rng.random_sample() < prob:
                    img, kp = fn(img, kp)
            out_imgs.append(np.ascontiguousarray(img))
            out_kps.append([tuple(v) for v in kp.tolist()])
        return np.stack(out_imgs), out_kps


def build_aug(apply_prob: float = 0.5, seed=None, backend: str = "auto"):
    """The reference's pipeline (fitdgp_util.py:412-436).  backend: 'imgaug' | 'numpy' | 'auto' (imgaug if importable)."""
    if backend in ("auto", "imgaug"):
        try:
            import imgaug.augmenters as iaa
        except ImportError:
            if backend == "imgaug":
                raise
        else:
            sometimes = lambda aug: iaa.Sometimes(apply_prob, aug)
            pipeline = iaa.Sequential(random_order=False)
            pipeline.add(sometimes(iaa.Fliplr(0.5)))
            pipeline.add(sometimes(iaa.Affine(rotate=(-10, 10))))
            pipeline.add(sometimes(iaa.MotionBlur(k=3, angle=(-90, 90))))
            pipeline.add(sometimes(iaa.CoarseDropout((0, 0.02), size_percent=(0.01, 0.05))))
            pipeline.add(sometimes(iaa.ElasticTransformation(sigma=5, alpha=(0, 10))))
            pipeline.add(sometimes(iaa.AdditiveGaussianNoise(loc=0, scale=(0.0, 0.01 * 255), per_channel=0.5)))
            pipeline.add(iaa.Sometimes(0.4, iaa.CropAndPad(percent=(-0.3, 0.1), keep_size=True)))
            return pipeline
    return NumpyAugPipeline(apply_prob, seed)


def data_aug(all_data_batch, visible_frame_within_batch, joint_loc, pipeline, dgp_cfg):
    """fitdgp_util.py:439-451: augment the visible frames; labels (row, col) in scoremap cells <-> (x, y) px."""
    stride = float(dgp_cfg.stride)
    visible_data = all_data_batch[visible_frame_within_batch, :, :, :].astype(np.uint8)
    xy = np.flip(np.asarray(joint_loc, dtype=np.float64), 2) * stride + stride / 2
    kp_list = [[tuple(f) for f in v.tolist()] for v in xy]
    batch_images, batch_joints = pipeline(images=visible_data, keypoints=kp_list)
    joint_loc_aug = np.flip(np.array(batch_joints, dtype=np.float64) / stride - 0.5, 2)
    all_data_batch_aug = np.copy(all_data_batch)
    all_data_batch_aug[visible_frame_within_batch, :, :, :] = np.asarray(batch_images)
    return all_data_batch_aug, joint_loc_aug
