"""Frame sources.  Video DECODE is out of scope (the reference uses moviepy's VideoFileClip,
DGP/models/eval.py:256, and this path leaves it untouched): a real video is opened through moviepy
when it is installed; directories of images, .npy/.npz stacks and in-memory arrays are accepted so the
pipeline also runs where no decoder exists (e.g. the bundled demo project ships only labeled PNGs)."""
from __future__ import annotations

import glob
import os
from typing import Iterator, Tuple

import numpy as np


class ArraySource:
    def __init__(self, frames: np.ndarray, fps: float = 30.0):
        assert frames.ndim == 4 and frames.shape[-1] == 3
        self.frames, self.fps = frames, fps
        self.n_frames = frames.shape[0]
        self.size = (frames.shape[2], frames.shape[1])          # (width, height) like VideoFileClip.size

    def iter_frames(self) -> Iterator[np.ndarray]:
        for f in self.frames:
            yield f

    def iter_batches(self, n: int) -> Iterator[np.ndarray]:
        """Contiguous runs of up to n frames (views): lets the staging thread fill a pinned batch with one copy."""
        for i in range(0, self.n_frames, n):
            yield self.frames[i:i + n]

    def close(self):
        pass


class ImageDirSource:
    def __init__(self, path: str, fps: float = 30.0):
        from PIL import Image
        self._Image = Image
        self.files = sorted(f for ext in ("png", "jpg", "jpeg", "bmp") for f in glob.glob(os.path.join(path, "*." + ext)))
        if not self.files:
            raise FileNotFoundError("no image frames in %s" % path)
        self.fps, self.n_frames = fps, len(self.files)
        with Image.open(self.files[0]) as im:
            self.size = im.size

    def iter_frames(self):
        for f in self.files:
            with self._Image.open(f) as im:
                yield np.asarray(im.convert("RGB"))

    def close(self):
        pass


class MoviepySource:
    def __init__(self, path: str):
        from moviepy.editor import VideoFileClip           # third-party decode, untouched
        self.clip = VideoFileClip(str(path))
        self.fps = self.clip.fps
        self.n_frames = int(np.ceil(self.clip.fps * self.clip.duration))     # eval.py:257
        self.size = tuple(self.clip.size)

    def iter_frames(self):
        return self.clip.iter_frames()

    def close(self):
        self.clip.close()


def open_frame_source(video_file):
    if isinstance(video_file, np.ndarray):
        return ArraySource(video_file)
    p = str(video_file)
    if os.path.isdir(p):
        return ImageDirSource(p)
    if p.endswith(".npy"):
        return ArraySource(np.load(p))
    if p.endswith(".npz"):
        z = np.load(p)
        return ArraySource(z[z.files[0]])
    if not os.path.exists(p):
        raise FileNotFoundError(p)
    try:
        return MoviepySource(p)
    except ImportError as e:
        raise ImportError("decoding %s needs moviepy (video decode is not part of this package); pass a directory "
                          "of frames or a .npy stack instead" % p) from e
