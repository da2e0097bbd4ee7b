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

    def frame_at(self, i: int) -> np.ndarray:
        return self.frames[int(i)]

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

    def _read(self, f):
        with self._Image.open(f) as im:
            return np.asarray(im.convert("RGB"))

    def frame_at(self, i: int) -> np.ndarray:
        return self._read(self.files[int(i)])

    def iter_frames(self):
        for f in self.files:
            yield self._read(f)

    def close(self):
        pass


class LabeledDirSource(ImageDirSource):
    """Pseudo-video from a DLC `labeled-data/<video>/` folder (img<NNN>.png = frame NNN of the video): n_frames = last
    labeled frame number + 1 and frame t shows the labeled image with the largest number <= t (the first one before that).
    Labeled frame NNN is therefore exactly img<NNN>.png, so the project's labels stay aligned.  Plumbing only -- the bundled
    Reaching demo project ships its 55 labeled PNGs but not its video (BASELINE configs[0])."""

    def __init__(self, path: str, fps: float = 30.0):
        super().__init__(path, fps)
        import re
        nums = []
        for f in self.files:
            m = re.match(r"img(\d+)\.", os.path.basename(f))
            if m:
                nums.append((int(m.group(1)), f))
        if not nums:
            raise FileNotFoundError("no img<NNN> frames in %s" % path)
        nums.sort()
        self.numbers = np.array([n for n, _ in nums])
        self.files = [f for _, f in nums]
        self.n_frames = int(self.numbers[-1]) + 1
        # a video has ONE frame size: the most common size of the labeled images (the Reaching project mixes 832 x 747 frames with
        # 640 x 470 crops); images of another size are resized to it
        from collections import Counter
        sizes = []
        for f in self.files:
            with self._Image.open(f) as im:
                sizes.append(im.size)
        self.size = Counter(sizes).most_common(1)[0][0]

    def _read(self, f):
        with self._Image.open(f) as im:
            im = im.convert("RGB")
            if im.size != self.size:
                im = im.resize(self.size, self._Image.BILINEAR)
            return np.asarray(im)

    def _file_for(self, t: int) -> str:
        k = int(np.searchsorted(self.numbers, t, side="right")) - 1
        return self.files[max(k, 0)]

    def frame_at(self, i: int) -> np.ndarray:
        return self._read(self._file_for(int(i)))

    def iter_frames(self):
        last, img = None, None
        for t in range(self.n_frames):
            f = self._file_for(t)
            if f != last:
                img, last = self._read(f), f
            yield img


class MoviepySource:
    def __init__(self, path: str):
        from moviepy.editor import VideoFileClip           # third-party decode, untouched
        self.clip = VideoFileClip(str(path))
        self.fps = self.clip.fps
        self.n_frames = int(np.ceil(self.clip.fps * self.clip.duration))     # eval.py:257
        self.size = tuple(self.clip.size)

    def frame_at(self, i: int) -> np.ndarray:
        return np.asarray(self.clip.get_frame(int(i) * 1.0 / self.clip.fps))      # seek by time like DGP/dataset.py:811-821

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
        # <proj>/videos/<name>.avi missing but <proj>/labeled-data/<name>/ present: the labeled frames as a pseudo-video
        stem = os.path.basename(p).rsplit(".", 1)[0]
        cand = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(p))), "labeled-data", stem)
        if os.path.isdir(cand):
            print("video %s is missing: using the labeled frames in %s as a pseudo-video" % (p, cand), flush=True)
            return LabeledDirSource(cand)
        raise FileNotFoundError(p)
    try:
        return MoviepySource(p)
    except ImportError as e:
        raise ImportError("decoding %s needs moviepy (video decode is not part of this package); pass a directory "
                          "of frames or a .npy stack instead" % p) from e
