"""The TF-session-shaped boundary of the training step.

The reference's training path sits behind `sess.run([loss, train_op], feed_dict)` (DGP/models/fitdgp.py:801-818) with the
12 placeholders returned by `dgp_loss` (:1130-1142) and a train_op built from MomentumOptimizer + clip_by_global_norm
(:708-713).  Callers written against that contract keep working: `dgp_loss` returns (loss, total_loss, total_loss_visible,
placeholders) whose members are the handles below, `TrainSession.run` evaluates them on the HIP engine
(Trainer.forward_backward / apply_gradients).  There is no graph: a handle is a name, `run` is one call into the C-ABI.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

PLACEHOLDER_KEYS = ("inputs", "targets", "locref_map", "locref_mask", "visible_marker_pl", "hidden_marker_pl",
                    "visible_marker_in_targets_pl", "wt_batch_mask_pl", "vector_field_tf", "nt_batch_pl", "wt_batch_pl",
                    "alpha_tf")                                   # fitdgp.py:1130-1142, same keys, same order
LOSS_KEYS = ("visible_loss_pred", "hidden_loss_pred", "visible_loss_locref", "ws_loss", "wt_loss", "total_loss")


class Placeholder:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "<dgp placeholder %s>" % self.name


class LossTensor:
    """An evaluable scalar of the loss graph (an entry of the loss dict, total_loss or total_loss_visible)."""

    def __init__(self, name, graph):
        self.name, self.graph = name, graph

    def __repr__(self):
        return "<dgp loss %s>" % self.name


class LossDict(dict):
    """`loss` of dgp_loss: name -> LossTensor; fetching the dict returns name -> float like sess.run on a dict of tensors."""
    graph = None


class TrainOp:
    def __init__(self, graph, objective: LossTensor, learning_rate, momentum, clip_norm):
        self.graph, self.objective, self.learning_rate = graph, objective, learning_rate
        self.momentum, self.clip_norm = momentum, clip_norm


class LossGraph:
    """What dgp_loss pre-computes (fitdgp.py:865-892) plus the handles it hands out."""

    def __init__(self, hyper, S0, ws, ws_max, n_frames_total, n_visible_frames_total, nj):
        self.hyper, self.S0, self.ws, self.ws_max = hyper, S0, ws, ws_max
        self.n_frames_total, self.n_visible_frames_total, self.nj = n_frames_total, n_visible_frames_total, nj
        self.placeholders: Dict[str, Placeholder] = {k: Placeholder(k) for k in PLACEHOLDER_KEYS}
        self.loss = LossDict({k: LossTensor(k, self) for k in LOSS_KEYS})
        self.loss.graph = self
        self.total_loss = self.loss["total_loss"]
        self.total_loss_visible = LossTensor("total_loss_visible", self)

    def minimize(self, objective: LossTensor, learning_rate, momentum: float = 0.9, clip_norm: float = 10.0) -> TrainOp:
        """MomentumOptimizer(learning_rate, 0.9) on clip_by_global_norm(gradients, 10) of `objective` w.r.t. every trainable
        (fitdgp.py:708-713; :414-418 for total_loss_visible).  learning_rate: a float or a Placeholder fed at run time."""
        if objective.name not in ("total_loss", "total_loss_visible"):
            raise ValueError("only total_loss / total_loss_visible can be minimised (fitdgp.py:416, 710)")
        return TrainOp(self, objective, learning_rate, momentum, clip_norm)


class TrainSession:
    """sess.run for the training graph.  One run = one forward (+ loss, + backward) on the Trainer; a fetched TrainOp also
    applies the update (and, data-parallel, averages the gradients over the ranks first)."""

    def __init__(self, trainer, graph: LossGraph):
        self.trainer, self.graph = trainer, graph

    def _batch(self, feed):
        ph = self.graph.placeholders
        need = ("inputs", "targets", "locref_map", "locref_mask", "visible_marker_pl", "hidden_marker_pl",
                "visible_marker_in_targets_pl")
        for k in need:
            if ph[k] not in feed:
                raise KeyError("feed_dict lacks placeholder %r" % k)
        images = feed[ph["inputs"]]
        if not (hasattr(images, "is_cuda") and images.is_cuda):      # (a device tensor: frames the fit drivers uploaded ahead of time)
            images = np.asarray(images)
        nt = images.shape[0]
        if ph["nt_batch_pl"] in feed and int(feed[ph["nt_batch_pl"]]) != nt:
            raise ValueError("nt_batch_pl = %s but inputs hold %d frames" % (feed[ph["nt_batch_pl"]], nt))
        batch = dict(targets=np.asarray(feed[ph["targets"]], dtype=np.float64),
                     locref_map=feed[ph["locref_map"]], locref_mask=feed[ph["locref_mask"]],
                     visible_marker=feed[ph["visible_marker_pl"]], hidden_marker=feed[ph["hidden_marker_pl"]],
                     visible_marker_in_targets=feed[ph["visible_marker_in_targets_pl"]])
        if ph["alpha_tf"] in feed:                 # the (row, col) grid: generated inside the kernels, checked here
            a = np.asarray(feed[ph["alpha_tf"]])
            lm = np.asarray(batch["locref_map"])
            if a.ndim != 3 or a.shape[0] != 2 or (lm.ndim == 4 and tuple(a.shape[1:]) != tuple(lm.shape[1:3])):
                raise ValueError("alpha_tf must be [2, nx_out, ny_out]")
        hy = self.graph.hyper
        if hy.wt > 0 and ph["vector_field_tf"] in feed and feed[ph["vector_field_tf"]] is not None:
            batch["vector_field"] = feed[ph["vector_field_tf"]]
            mask = np.asarray(feed.get(ph["wt_batch_mask_pl"], np.ones(max(nt - 1, 0))), dtype=np.float32)
            wtb = np.asarray(feed.get(ph["wt_batch_pl"], np.ones(max(nt - 1, 0)) * hy.wt), dtype=np.float32)
            batch["wt_batch_mask"] = mask * wtb / max(hy.wt, 1e-30)         # the kernels take wt * mask (fitdgp.py:905)
        return images, batch

    def run(self, fetches, feed_dict):
        from .fitdgp import _frames_to_device
        single = not isinstance(fetches, (list, tuple))
        fl = [fetches] if single else list(fetches)
        train_ops = [f for f in fl if isinstance(f, TrainOp)]
        if len(train_ops) > 1:
            raise ValueError("one train_op per run")
        images, batch = self._batch(feed_dict)
        g, tr = self.graph, self.trainer
        labeled_only = bool(train_ops) and train_ops[0].objective.name == "total_loss_visible"
        if not train_ops and all(isinstance(f, LossTensor) and f.name == "total_loss_visible" for f in fl):
            labeled_only = True
        frames = _frames_to_device(tr, images)
        losses = tr.forward_backward(frames, batch, g.hyper, g.S0, g.ws, g.ws_max, g.n_frames_total, g.n_visible_frames_total,
                                     labeled_only=labeled_only)
        if train_ops:
            op = train_ops[0]
            lr = op.learning_rate
            if isinstance(lr, Placeholder):
                if lr not in feed_dict:
                    raise KeyError("feed_dict lacks the learning-rate placeholder")
                lr = feed_dict[lr]
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                tr.allreduce_gradients()
            losses["grad_norm"] = tr.apply_gradients(float(lr), op.momentum, op.clip_norm)
        out = []
        for f in fl:
            if isinstance(f, TrainOp):
                out.append(None)
            elif isinstance(f, LossDict):
                out.append({k: losses.get(k, 0.0) for k in list(f.keys()) + (["grad_norm"] if "grad_norm" in losses else [])})
            elif isinstance(f, LossTensor):
                out.append(losses[f.name])
            else:
                raise TypeError("cannot fetch %r" % (f,))
        return out[0] if single else out
