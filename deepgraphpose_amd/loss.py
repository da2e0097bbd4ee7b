"""Torch-facing wrapper of the DGP loss kernels (include/dgp_hip.h: dgp_loss_fwd_bwd).

`dgp_loss_fwd_bwd` evaluates the loss terms of dgp_loss (DGP/models/fitdgp.py:946-1076) on the head outputs
and returns d total_loss / d pred and d total_loss / d locref_pred; the backbone backward consumes those."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from . import _lib
from .engine import _ptr, _stream, _need_cuda

LOSS_NAMES = ("visible_loss_pred", "hidden_loss_pred", "visible_loss_locref", "ws_loss", "total_loss",
              "total_loss_visible", "wt_loss")


@dataclass
class DGPHyper:
    """Hyper-parameters fit_dgp hard-codes on the config singleton (fitdgp.py:637-654), made explicit."""
    ws: float = 1000.0
    ws_max: float = 1.2
    wt: float = 0.0
    wt_max: float = 0.0
    wn_visible: float = 5.0
    wn_hidden: float = 3.0
    gamma: float = 1.0
    gauss_len: int = 1
    lengthscale: float = 1.0
    lr: float = 0.005
    gm2: int = 0
    gm3: int = 0
    stride: float = 8.0
    locref_loss_weight: float = 0.05
    locref_huber_loss: bool = True
    momentum: float = 0.9
    clip_norm: float = 10.0


def _dev_i32(a, dev):
    return torch.as_tensor(np.asarray(a, dtype=np.int32), device=dev).contiguous()


def _dev_f32(a, dev):
    return torch.as_tensor(np.asarray(a, dtype=np.float32), device=dev).contiguous()


class LossInputs:
    """Device-resident inputs of one loss evaluation (dgp_loss_prepare): everything that does not depend on the network's output."""
    __slots__ = ("desc", "dev", "shape", "targets", "lmap", "lmask", "vm", "hm", "vt", "S0", "ws", "ws_max", "vf", "wtb", "scratch")


def dgp_loss_prepare(nt: int, H: int, W: int, nj: int, batch: dict, hyper: DGPHyper, S0, ws, ws_max, n_frames_total: float,
                     n_visible_frames_total: float, dev) -> LossInputs:
    """Host validation + upload of the batch's index / target arrays.  Independent of the forward pass, so the trainer calls it
    BEFORE launching the forward: the loss kernels then follow the forward on the stream without the GPU waiting for the host."""
    lib = _lib.load()
    # the kernels use these arrays as indices: check them here, on the host, before anything is launched
    vm_h = np.asarray(batch["visible_marker"], dtype=np.int64).ravel()
    hm_h = np.asarray(batch["hidden_marker"], dtype=np.int64).ravel()
    vt_h = np.asarray(batch["visible_marker_in_targets"], dtype=np.int64).ravel()
    tg_h = np.nan_to_num(np.asarray(batch["targets"], dtype=np.float64), nan=0.0).reshape(-1, 2)
    for name, a, hi in (("visible_marker", vm_h, nt * nj), ("hidden_marker", hm_h, nt * nj),
                        ("visible_marker_in_targets", vt_h, tg_h.shape[0])):
        if a.size and (a.min() < 0 or a.max() >= hi):
            raise ValueError("%s out of range [0, %d): min %d max %d" % (name, hi, a.min(), a.max()))
    if vt_h.size != vm_h.size:
        raise ValueError("visible_marker_in_targets has %d entries for %d visible markers" % (vt_h.size, vm_h.size))
    if not 1 <= int(hyper.gauss_len) <= 7:
        raise ValueError("gauss_len must be 1..7")
    li = LossInputs()
    li.dev, li.shape = dev, (nt, H, W, nj)
    li.vm, li.hm, li.vt = _dev_i32(vm_h, dev), _dev_i32(hm_h, dev), _dev_i32(vt_h, dev)
    li.targets = _dev_f32(tg_h, dev)
    li.lmap, li.lmask = _dev_f32(batch["locref_map"], dev), _dev_f32(batch["locref_mask"], dev)
    for name, t in (("locref_map", li.lmap), ("locref_mask", li.lmask)):
        if vm_h.size and tuple(t.shape) != (nt, H, W, 2 * nj):
            raise ValueError("%s %s does not match the prediction grid %s" % (name, tuple(t.shape), (nt, H, W, 2 * nj)))
    S0 = np.asarray(S0, dtype=np.float32).reshape(-1, nj)
    nl = S0.shape[0]
    li.S0, li.ws, li.ws_max = _dev_f32(S0, dev), _dev_f32(ws, dev), _dev_f32(ws_max, dev)
    use_wt = hyper.wt > 0 and nt > 1 and batch.get("vector_field") is not None
    li.vf = li.wtb = None
    hin = win = 0
    if use_wt:          # temporal clique: flow magnitude [nt-1,Hin,Win] (learn_wt) and wt * batch_mask (fitdgp.py:774,905)
        li.vf = _dev_f32(batch["vector_field"], dev)
        hin, win = int(li.vf.shape[1]), int(li.vf.shape[2])
        mask = np.asarray(batch.get("wt_batch_mask", np.ones(nt - 1)), dtype=np.float32)
        li.wtb = _dev_f32(np.ones(nt - 1, dtype=np.float32) * hyper.wt * mask, dev)
    li.desc = _lib.DgpLossDesc(nt, H, W, nj, nl, li.vm.numel(), li.hm.numel(), hyper.gm2, hyper.gm3, hyper.gauss_len,
                               int(hyper.locref_huber_loss), hyper.gamma, hyper.lengthscale, hyper.stride, hyper.wn_visible,
                               hyper.wn_hidden, hyper.locref_loss_weight, float(n_frames_total), float(n_visible_frames_total),
                               int(use_wt), hin, win, float(hyper.wt_max))
    nb = C.c_size_t()
    _lib.check(lib.dgp_loss_scratch_bytes(C.byref(li.desc), C.byref(nb)), "dgp_loss_scratch_bytes")
    li.scratch = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    return li


def dgp_loss_launch(li: LossInputs, pred: torch.Tensor, locref_pred: torch.Tensor):
    """Enqueue the loss forward + backward kernels on the current stream: -> (losses [8] DEVICE tensor in LOSS_NAMES order, dpred,
    dlocref, mu [nt,nj,2]).  No host synchronisation."""
    lib = _lib.load()
    _need_cuda(pred, torch.float32, "pred")
    _need_cuda(locref_pred, torch.float32, "locref_pred")
    nt, H, W, nj = li.shape
    if tuple(pred.shape) != (nt, H, W, nj):
        raise ValueError("pred %s does not match the prepared batch %s" % (tuple(pred.shape), (nt, H, W, nj)))
    if tuple(locref_pred.shape) != (nt, H, W, 2 * nj):
        raise ValueError("locref_pred %s does not match pred %s" % (tuple(locref_pred.shape), tuple(pred.shape)))
    dev = pred.device
    dpred, dloc = torch.empty_like(pred), torch.empty_like(locref_pred)
    mu = torch.empty((nt, nj, 2), dtype=torch.float32, device=dev)
    losses = torch.zeros(8, dtype=torch.float32, device=dev)
    _lib.check(lib.dgp_loss_fwd_bwd(C.byref(li.desc), _ptr(pred), _ptr(locref_pred), _ptr(li.targets), _ptr(li.lmap), _ptr(li.lmask),
                                    _ptr(li.vm), _ptr(li.hm), _ptr(li.vt), _ptr(li.S0), _ptr(li.ws), _ptr(li.ws_max), _ptr(li.vf),
                                    _ptr(li.wtb), _ptr(dpred), _ptr(dloc), _ptr(mu), _ptr(losses), _ptr(li.scratch), li.scratch.numel(),
                                    _stream(dev)), "dgp_loss_fwd_bwd")
    return losses, dpred, dloc, mu


def losses_to_dict(losses: torch.Tensor) -> dict:
    lv = losses.cpu().numpy()
    return {k: float(lv[i]) for i, k in enumerate(LOSS_NAMES)}


def dgp_loss_fwd_bwd(pred: torch.Tensor, locref_pred: torch.Tensor, batch: dict, hyper: DGPHyper, S0, ws, ws_max,
                     n_frames_total: float, n_visible_frames_total: float):
    """pred [nt,H,W,nj], locref_pred [nt,H,W,2nj] device fp32.  batch: targets [nv,nj,2] (NaN = unlabeled),
    locref_map / locref_mask [nt,H,W,2nj], visible_marker, hidden_marker, visible_marker_in_targets.
    -> (losses dict of python floats, dpred, dlocref, mu [nt,nj,2]).  (= dgp_loss_prepare + dgp_loss_launch + a read-back.)"""
    _need_cuda(pred, torch.float32, "pred")
    _need_cuda(locref_pred, torch.float32, "locref_pred")
    nt, H, W, nj = pred.shape
    if tuple(locref_pred.shape) != (nt, H, W, 2 * nj):
        raise ValueError("locref_pred %s does not match pred %s" % (tuple(locref_pred.shape), tuple(pred.shape)))
    li = dgp_loss_prepare(nt, H, W, nj, batch, hyper, S0, ws, ws_max, n_frames_total, n_visible_frames_total, pred.device)
    losses, dpred, dloc, mu = dgp_loss_launch(li, pred, locref_pred)
    return losses_to_dict(losses), dpred, dloc, mu
