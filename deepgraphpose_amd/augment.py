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
        if seed is None:        # reproducible under np.random.seed(...) like the rest of the fit drivers' schedule (not OS entropy)
            seed = int(np.random.randint(0, 2 ** 31 - 1))
        self.rng = np.random.RandomState(seed)
        self.bulk = np.random.Generator(np.random.PCG64(int(self.rng.randint(0, 2 ** 31 - 1))))      # per-pixel noise fields
        # the elastic warp runs on torch's CPU thread pool, from the fit drivers' prefetch thread, next to the thread that feeds the
        # GPU: on a 256-core host an unbounded pool (and its spinning idle workers) halves the launch rate of the training step.
        # A frame needs no more than a few cores.
        try:
            import torch
            if torch.get_num_threads() > 8:
                torch.set_num_threads(8)
        except ImportError:
            pass

    # ---- the seven augmenters: (image uint8 [H,W,C], keypoints float [nj,2] as (x, y)) -> the same ----
    def fliplr(self, img, kp):
        if self.rng.random_sample() < 0.5:
            img = img[:, ::-1]
            kp = kp.copy()
            kp[:, 0] = img.shape[1] - kp[:, 0]
        return img, kp

    def rotate(self, img, kp):
        from PIL import Image
        deg = self.rng.uniform(-10, 10)
        t = np.deg2rad(deg)
        H, W = img.shape[:2]
        c = np.array([(W - 1) / 2.0, (H - 1) / 2.0])                      # rotation about the image centre, (x, y)
        R = np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]])   # output = R (input - c) + c in (x, y)
        Ri = R.T                                                          # inverse map for the resampler: in = Ri (out - c) + c
        off = c - Ri @ c
        # PIL samples the input at a (xo + 0.5) + b (yo + 0.5) + c0 - 0.5 (pixel-centre convention); fold the halves into c0 / f0 so
        # that pixel INDICES map as in = Ri out + off (bilinear, zeros outside -- imgaug's Affine defaults: order 1, cval 0)
        a_, b_, d_, e_ = Ri[0, 0], Ri[0, 1], Ri[1, 0], Ri[1, 1]
        c0 = off[0] + 0.5 - 0.5 * (a_ + b_)
        f0 = off[1] + 0.5 - 0.5 * (d_ + e_)
        out = np.asarray(Image.fromarray(img).transform((W, H), Image.AFFINE, (a_, b_, c0, d_, e_, f0), resample=Image.BILINEAR))
        kp2 = (kp - c) @ R.T + c
        return out, kp2

    def motion_blur(self, img, kp):
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
        # 3 x 3 convolution (correlation with the flipped kernel), borders repeat the edge pixel (scipy's "reflect" at pad 1), all
        # channels at once: a sum of the shifted images with non-zero weight
        H, W = img.shape[:2]
        pad = np.pad(img, ((1, 1), (1, 1), (0, 0)), mode="edge").astype(np.float32)
        out = np.zeros(img.shape, dtype=np.float32)
        for yy in range(3):
            for xx in range(3):
                wgt = k[2 - yy, 2 - xx]
                if wgt != 0.0:
                    out += np.float32(wgt) * pad[yy:yy + H, xx:xx + W]
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
        """ElasticTransformation(sigma=5, alpha=(0, 10)): out[p] = in[p + d(p)], d = alpha * gaussian_filter(uniform(-1, 1), sigma).
        With sigma = 5 the field is smooth over tens of pixels, so it is synthesised on a grid of every 4th pixel -- white noise of
        variance (1/3) / 16 filtered with sigma / 4: the same variance and correlation length as the full-resolution field -- and
        brought to full resolution bilinearly; the resampling (bilinear, zeros outside: imgaug's order 1 / cval 0) is torch's
        multi-threaded CPU grid_sample instead of one scipy map_coordinates per channel (100 -> 8 ms on a 640 x 480 frame)."""
        import torch
        import torch.nn.functional as F
        from scipy import ndimage
        H, W = img.shape[:2]
        alpha = self.rng.uniform(0, 10)
        S = 4
        hc, wc = -(-(H - 1) // S) + 1, -(-(W - 1) // S) + 1               # node (i, j) sits on pixel (S i, S j); the last node at or past the edge
        dxc = ndimage.gaussian_filter((self.rng.random_sample((hc, wc)) * 2 - 1) / S, 5.0 / S, mode="constant") * alpha
        dyc = ndimage.gaussian_filter((self.rng.random_sample((hc, wc)) * 2 - 1) / S, 5.0 / S, mode="constant") * alpha
        d = torch.from_numpy(np.stack([dxc, dyc]).astype(np.float32))[None]                     # [1, 2, hc, wc]
        d = F.interpolate(d, size=(S * (hc - 1) + 1, S * (wc - 1) + 1), mode="bilinear", align_corners=True)[0, :, :H, :W]
        yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        gx = (xx + d[0]) * (2.0 / max(W - 1, 1)) - 1.0
        gy = (yy + d[1]) * (2.0 / max(H - 1, 1)) - 1.0
        src = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1)[None].float()
        out = F.grid_sample(src, torch.stack([gx, gy], -1)[None], mode="bilinear", padding_mode="zeros", align_corners=True)
        out = out[0].permute(1, 2, 0).round_().clamp_(0, 255).to(torch.uint8).numpy()
        kp2 = kp.copy()
        dn = d.numpy()
        for j in range(kp.shape[0]):                                      # out[p] = in[p + d(p)]  =>  a point moves by about -d
            x, y = kp[j]
            if np.isfinite(x) and np.isfinite(y):
                xi, yi = int(np.clip(round(x), 0, W - 1)), int(np.clip(round(y), 0, H - 1))
                kp2[j, 0] = x - dn[0, yi, xi]
                kp2[j, 1] = y - dn[1, yi, xi]
        return out, kp2

    def gaussian_noise(self, img, kp):
        scale = self.rng.uniform(0.0, 0.01 * 255)
        per_channel = self.rng.random_sample() < 0.5
        shape = img.shape if per_channel else img.shape[:2] + (1,)
        noise = self.bulk.standard_normal(size=shape, dtype=np.float32)   # (the bulk stream: 4x faster than RandomState.normal)
        out = img.astype(np.float32) + np.float32(scale) * noise
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
