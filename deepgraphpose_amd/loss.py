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


class _PackedUpload:
    """Several host arrays -> ONE pinned staging buffer -> ONE asynchronous host-to-device copy on the current stream; the arrays come
    back as typed views of one device buffer.  (Nine separate uploads from pageable memory cost the training step ~0.5 ms of idle GPU:
    each is a blocking staged copy.)  Two staging buffers alternate; a buffer is reused only after the copy that read it has finished."""
    ALIGN = 256

    def __init__(self):
        self._host = [None, None]
        self._done = [None, None]
        self._k = 0

    def __call__(self, arrays, dev):
        offs, total = [], 0
        for a in arrays:
            offs.append(total)
            total += (a.nbytes + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        total = max(total, self.ALIGN)
        k = self._k = self._k ^ 1
        if self._done[k] is not None:
            self._done[k].synchronize()
        if self._host[k] is None or self._host[k].numel() < total:
            self._host[k] = torch.empty(total * 2, dtype=torch.uint8).pin_memory()
        host = self._host[k]
        hview = host.numpy()
        for a, o in zip(arrays, offs):
            if a.nbytes:
                hview[o:o + a.nbytes] = np.ascontiguousarray(a).reshape(-1).view(np.uint8)
        devbuf = torch.empty(total, dtype=torch.uint8, device=dev)
        devbuf.copy_(host[:total], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._done[k] = ev
        out = []
        for a, o in zip(arrays, offs):
            t = devbuf[o:o + a.nbytes].view(torch.int32 if a.dtype == np.int32 else torch.float32).view(a.shape)
            out.append(t)
        return out


_packed_upload = {}


def _upload_all(arrays, dev):
    key = str(dev)
    if key not in _packed_upload:
        _packed_upload[key] = _PackedUpload()
    return _packed_upload[key](arrays, dev)


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
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    lmap_h, lmask_h = f32(batch["locref_map"]), f32(batch["locref_mask"])
    for name, t in (("locref_map", lmap_h), ("locref_mask", lmask_h)):
        if vm_h.size and tuple(t.shape) != (nt, H, W, 2 * nj):
            raise ValueError("%s %s does not match the prediction grid %s" % (name, tuple(t.shape), (nt, H, W, 2 * nj)))
    S0 = f32(S0).reshape(-1, nj)
    nl = S0.shape[0]
    use_wt = hyper.wt > 0 and nt > 1 and batch.get("vector_field") is not None
    host = [vm_h.astype(np.int32), hm_h.astype(np.int32), vt_h.astype(np.int32), tg_h.astype(np.float32), lmap_h, lmask_h, S0, np.atleast_1d(f32(ws)), np.atleast_1d(f32(ws_max))]
    hin = win = 0
    if use_wt:          # temporal clique: flow magnitude [nt-1,Hin,Win] (learn_wt) and wt * batch_mask (fitdgp.py:774,905)
        vf_h = f32(batch["vector_field"])
        hin, win = int(vf_h.shape[1]), int(vf_h.shape[2])
        mask = np.asarray(batch.get("wt_batch_mask", np.ones(nt - 1)), dtype=np.float32)
        host += [vf_h, np.ones(nt - 1, dtype=np.float32) * hyper.wt * mask]
    up = _upload_all(host, dev)          # one staging buffer, one asynchronous copy
    li.vm, li.hm, li.vt, li.targets, li.lmap, li.lmask, li.S0, li.ws, li.ws_max = up[:9]
    li.vf, li.wtb = (up[9], up[10]) if use_wt else (None, None)
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
