"""Optional annotated-movie writer (DGP/models/eval.py:46-119 draws with matplotlib + moviepy).
Rendering is outside the hot path; this thin version needs moviepy and draws square dots with numpy."""
from __future__ import annotations

import numpy as np


def create_annotated_movie(video_file, x, y, mask_array=None, filename="movie.mp4", dotsize=3, colormap="jet"):
    from moviepy.editor import VideoFileClip
    clip = VideoFileClip(str(video_file))
    nj, T = x.shape
    if mask_array is None:
        mask_array = ~np.isnan(x)
    palette = (np.stack([np.linspace(0, 255, nj), np.linspace(255, 0, nj), np.full(nj, 128)], 1)).astype(np.uint8)
    fps = clip.fps

    def draw(get_frame, t):
        img = get_frame(t).copy()
        i = min(int(round(t * fps)), T - 1)
        for j in range(nj):
            if mask_array[j, i]:
                r, c = int(round(y[j, i])), int(round(x[j, i]))
                img[max(r - dotsize, 0):r + dotsize + 1, max(c - dotsize, 0):c + dotsize + 1] = palette[j]
        return img

    clip.fl(draw).write_videofile(filename, fps=fps, codec="mpeg4", audio=False)
    clip.close()
    return filename
