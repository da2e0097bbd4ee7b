"""Snapshot I/O.  Weights are a dict {TF variable name: fp32 array in TF layout}; on disk a snapshot is
`<prefix>.npz` (numpy) or `<prefix>.safetensors`.  Reading TF-1 checkpoints (`.ckpt`, V2 `.index/.data`)
needs a bundle reader that is not built yet and raises a clear error."""
from __future__ import annotations

import os
from typing import Dict

import numpy as np


def resolve(path: str) -> str:
    p = str(path)
    for cand in (p, p + ".npz", p + ".safetensors"):
        if os.path.isfile(cand):
            return cand
    if os.path.isfile(p + ".index") or p.endswith(".ckpt"):
        raise NotImplementedError(
            "%s looks like a TensorFlow checkpoint; converting TF bundles needs the TF-checkpoint reader "
            "(planned, SURVEY.md 8(f) N1).  Export the variables to .npz with their TF names instead." % p)
    raise FileNotFoundError("snapshot not found: %s(.npz|.safetensors)" % p)


def load_weights(path: str) -> Dict[str, np.ndarray]:
    f = resolve(path)
    if f.endswith(".safetensors"):
        from safetensors.numpy import load_file
        return {k: np.asarray(v, dtype=np.float32) for k, v in load_file(f).items()}
    with np.load(f) as z:
        return {k: np.asarray(z[k], dtype=np.float32) for k in z.files}


def save_weights(path: str, weights: Dict[str, np.ndarray]) -> str:
    p = str(path)
    if not p.endswith(".npz"):
        p += ".npz"
    os.makedirs(os.path.dirname(os.path.abspath(p)), exist_ok=True)
    np.savez(p, **{k: np.asarray(v, dtype=np.float32) for k, v in weights.items()})
    return p


def net_depth(weights: Dict[str, np.ndarray]) -> int:
    for k in weights:
        if k.startswith("resnet_v1_"):
            return int(k.split("/")[0].split("_")[-1])
    raise KeyError("no resnet_v1_* variables in snapshot")
