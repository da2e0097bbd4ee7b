"""Snapshot I/O.  Weights are a dict {TF variable name: fp32 array in TF layout}.  On disk a snapshot is what the
reference's tf.train.Saver writes and restores (DGP/models/fitdgp.py:149-152,239-245,689-696,832-839; eval.py:194-211):
a TensorFlow-1 V2 bundle `<prefix>.index` + `<prefix>.data-00000-of-00001` -- the DEFAULT format of every fit driver --
or a single-file V1 `.ckpt` (the ImageNet resnet_v1_50.ckpt), both read and written by tf_checkpoint.py without
TensorFlow.  `<prefix>.npz` / `<prefix>.safetensors` remain as the opt-out (DGP_SNAPSHOT_FORMAT=npz)."""
from __future__ import annotations

import os
from typing import Dict

import numpy as np


SUFFIXES = (".index", ".npz", ".safetensors")


def default_format() -> str:
    """"tf" (V2 bundle, what the reference's Saver writes) unless DGP_SNAPSHOT_FORMAT=npz asks for the numpy opt-out."""
    fmt = os.environ.get("DGP_SNAPSHOT_FORMAT", "tf").lower()
    if fmt not in ("tf", "npz"):
        raise ValueError("DGP_SNAPSHOT_FORMAT must be 'tf' or 'npz', not %r" % fmt)
    return fmt


def exists(prefix: str) -> bool:
    """Is there a snapshot under this prefix, in any format (the fit drivers' skip-if-already-run guards)?"""
    p = str(prefix)
    return any(os.path.isfile(p + ext) for ext in SUFFIXES) or (os.path.isfile(p) and _is_table(p))


def resolve(path: str) -> str:
    p = str(path)
    if os.path.isfile(p + ".index"):                 # a V2 prefix wins: that is what a TF user passes (Saver.restore(sess, prefix))
        return p + ".index"
    for cand in (p, p + ".npz", p + ".safetensors"):
        if os.path.isfile(cand):
            return cand
    raise FileNotFoundError("snapshot not found: %s(.index|.npz|.safetensors)" % p)


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


def save_weights(path: str, weights: Dict[str, np.ndarray], fmt: str = None) -> str:
    p = str(path)
    if fmt is None:
        fmt = default_format()
    if fmt == "tf":
        from .tf_checkpoint import write_v2
        write_v2(p, weights)
        return p
    if not p.endswith(".npz"):
        p += ".npz"
    os.makedirs(os.path.dirname(os.path.abspath(p)), exist_ok=True)
    np.savez(p, **{k: np.asarray(v, dtype=np.float32) for k, v in weights.items()})
    return p


class Saver:
    """The part of tf.train.Saver the fit drivers use (DGP/models/fitdgp.py:150-152, 239-245, 696, 832-839):
    `save(weights, save_path, global_step)` writes `<save_path>-<global_step>` (TF joins them with a dash, so the
    reference's `snapshot-step2-` + 0 is `snapshot-step2--0`), keeps the directory's `checkpoint` state file
    (CheckpointState text proto: model_checkpoint_path + all_model_checkpoint_paths) and deletes the oldest
    snapshot once more than `max_to_keep` have been written by this Saver."""

    def __init__(self, max_to_keep: int = 5, fmt: str = None):
        self.max_to_keep = int(max_to_keep)
        self.fmt = default_format() if fmt is None else fmt
        self.kept = []

    @staticmethod
    def _files(prefix: str):
        return [prefix + ".index", prefix + ".data-00000-of-00001", prefix + ".npz"]

    def save(self, weights: Dict[str, np.ndarray], save_path: str, global_step=None) -> str:
        prefix = str(save_path) if global_step is None else "%s-%d" % (save_path, int(global_step))
        save_weights(prefix, weights, fmt=self.fmt)
        if prefix in self.kept:
            self.kept.remove(prefix)
        self.kept.append(prefix)
        while self.max_to_keep > 0 and len(self.kept) > self.max_to_keep:
            old = self.kept.pop(0)
            for f in self._files(old):
                if os.path.isfile(f):
                    os.remove(f)
        d = os.path.dirname(os.path.abspath(prefix))
        rel = [os.path.basename(k) if os.path.dirname(os.path.abspath(k)) == d else k for k in self.kept]
        with open(os.path.join(d, "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "%s"\n' % rel[-1])
            for r in rel:
                f.write('all_model_checkpoint_paths: "%s"\n' % r)
        return prefix


def latest_checkpoint(directory: str):
    """tf.train.latest_checkpoint: the prefix named by <directory>/checkpoint, or None."""
    state = os.path.join(str(directory), "checkpoint")
    if not os.path.isfile(state):
        return None
    with open(state) as f:
        for line in f:
            if line.startswith("model_checkpoint_path:"):
                name = line.split(":", 1)[1].strip().strip('"')
                return name if os.path.isabs(name) else os.path.join(str(directory), name)
    return None


def net_depth(weights: Dict[str, np.ndarray]) -> int:
    for k in weights:
        if k.startswith("resnet_v1_"):
            return int(k.split("/")[0].split("_")[-1])
    raise KeyError("no resnet_v1_* variables in snapshot")
