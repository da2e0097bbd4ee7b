"""Snapshot I/O.  Weights are a dict {TF variable name: fp32 array in TF layout}; on disk a snapshot is
`<prefix>.npz` (numpy), `<prefix>.safetensors`, or a TensorFlow-1 checkpoint -- a V2 bundle
(`<prefix>.index` + `<prefix>.data-*`) or a single-file V1 `.ckpt` -- read by tf_checkpoint.py without
TensorFlow.  `save_weights(..., fmt="tf")` writes a V2 bundle a TF Saver can restore."""
from __future__ import annotations

import os
from typing import Dict

import numpy as np


def resolve(path: str) -> str:
    p = str(path)
    for cand in (p, p + ".npz", p + ".safetensors"):
        if os.path.isfile(cand):
            return cand
    if os.path.isfile(p + ".index"):
        return p + ".index"
    raise FileNotFoundError("snapshot not found: %s(.npz|.safetensors|.index)" % p)


def _is_table(f: str) -> bool:
    from .tf_checkpoint import TABLE_MAGIC
    import struct
    try:
        with open(f, "rb") as fh:
            fh.seek(-8, os.SEEK_END)
            return struct.unpack("<Q", fh.read(8))[0] == TABLE_MAGIC
    except (OSError, struct.error):
        return False


def load_weights(path: str) -> Dict[str, np.ndarray]:
    f = resolve(path)
    if f.endswith(".index"):
        from .tf_checkpoint import load_checkpoint
        return load_checkpoint(f[:-len(".index")])
    if not f.endswith((".npz", ".safetensors")) and _is_table(f):
        from .tf_checkpoint import load_checkpoint
        return load_checkpoint(f)
    if f.endswith(".safetensors"):
        from safetensors.numpy import load_file
        return {k: np.asarray(v, dtype=np.float32) for k, v in load_file(f).items()}
    with np.load(f) as z:
        return {k: np.asarray(z[k], dtype=np.float32) for k in z.files}


def save_weights(path: str, weights: Dict[str, np.ndarray], fmt: str = "npz") -> str:
    p = str(path)
    if fmt == "tf":
        from .tf_checkpoint import write_v2
        write_v2(p, weights)
        return p
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
