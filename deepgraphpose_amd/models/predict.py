"""Host side of DLC's arg-max pose read-out (PET/nnet/predict.py:45-77).

The arg-max itself (sigmoid + first row-major maximum + locref gather) runs in the HIP kernel
`hard_argmax`; this module does the last float64 arithmetic exactly like the reference:
    pose = maxloc * stride + 0.5 * stride + locref[maxloc][j][::-1] * locref_stdev
"""
from __future__ import annotations

import numpy as np


def pose_from_argmax(idx, prob, offs, stride: float, locref_stdev: float = 7.2801) -> np.ndarray:
    """idx [C,2] (row, col) int, prob [C], offs [C,2] raw (dx, dy) locref or None -> pose [C,3] (x, y, p)."""
    idx = np.asarray(idx)
    pos = idx.astype("float") * stride + 0.5 * stride            # (row, col) -> (y, x) px, float64
    if offs is not None:
        off = (np.asarray(offs, dtype=np.float32) * np.float32(locref_stdev))   # fp32 like `locref *= stdev`
        pos = pos + off[:, ::-1]                                  # (dy, dx)
    return np.hstack((pos[:, ::-1], np.asarray(prob)[:, None].astype(np.float64)))


def argmax_pose_predict(scmap, offmat, stride):
    """Drop-in for PET/nnet/predict.py:62-77 on device tensors.

    scmap: torch fp32 [H,W,C] RAW logits on the GPU (the reference passes sigmoid(scmap); the kernel
    applies tf.sigmoid itself), offmat: torch fp32 [H,W,C,2] locref ALREADY scaled by locref_stdev, or None.
    """
    import torch
    from .. import engine
    loc = None if offmat is None else offmat.reshape(1, offmat.shape[0], offmat.shape[1], -1).contiguous()
    idx, prob, offs = engine.hard_argmax(scmap[None].contiguous(), loc)
    return pose_from_argmax(idx[0].cpu().numpy(), prob[0].cpu().numpy(),
                            None if offmat is None else offs[0].cpu().numpy(), stride, 1.0)
