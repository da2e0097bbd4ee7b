"""Training step on the MI355X engine: one `Trainer.step` = sess.run([loss, train_op])
of fit_dgp / fit_dgp_labeledonly (DGP/models/fitdgp.py:818, :505)."""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, engine
from .engine import _ptr, _stream
from .loss import DGPHyper, LOSS_NAMES


class Trainer:
    def __init__(self, depth: int, num_joints: int, in_h: int, in_w: int, max_frames: int = 11, device: int = 0,
                 tier: Optional[str] = None):
        """tier: None / "parity": fp32-class arithmetic, fp32 retained activations (the tier every parity test runs on); "f16": the 16-bit
        tier (include/dgp_hip.h, dgp_trainer_set_tier) -- BASELINE configs[3]'s precision: blocks 2-4 keep activations and gradient
        tensors as 2-byte H1 cells with scales predicted from the previous step, one MFMA per product, weight gradients by LDS-DMA on
        the tensors themselves; fp32 master weights / momentum / accumulation.  The first step of a shape (and any step whose tensors
        leave their predicted ranges) runs on the parity path, so callers never see an invalid step."""
        self.lib = _lib.load()
        self.net = engine.DGPNet(depth, num_joints, in_h, in_w, max_batch=max_frames, with_locref=True, device=device)
        self.device = self.net.device
        h = C.c_void_p()
        _lib.check(self.lib.dgp_trainer_create(self.net._h, C.byref(h)), "dgp_trainer_create")
        self._t = h
        nt, ntr, nst = C.c_int32(), C.c_int64(), C.c_int64()
        _lib.check(self.lib.dgp_trainer_num_tensors(h, C.byref(nt), C.byref(ntr), C.byref(nst)))
        self.n_trainable, self.n_stat = ntr.value, nst.value
        self.table = {}
        buf = C.create_string_buffer(256)
        for i in range(nt.value):
            off, size, st = C.c_int64(), C.c_int64(), C.c_int32()
            _lib.check(self.lib.dgp_trainer_tensor_info(h, i, buf, 256, C.byref(off), C.byref(size), C.byref(st)))
            self.table[buf.value.decode()] = (off.value, size.value, bool(st.value))
        self._ws = None
        self._ws_nt = 0
        self._shapes = {}
        self.tier = "parity"
        self.fast_redos = 0
        self.fast_passes = 0
        self._comm_stream = None
        if tier is not None:
            self.set_tier(tier)

    def set_tier(self, tier: str):
        t = {"parity": 0, "f16": 1}[tier]
        _lib.check(self.lib.dgp_trainer_set_tier(self._t, t), "dgp_trainer_set_tier")
        self.tier = tier
        self._fast_key = None                    # the next pass is a plain one (it leaves the ranges the predicted scales come from)
        if self.table and getattr(self, "_synced", False):
            self.sync()                          # (the 16-bit tier's weight cells are built by the sync)

    def __del__(self):
        t = getattr(self, "_t", None)
        if t:
            self.lib.dgp_trainer_destroy(t)
            self._t = None

    # ---- parameters ------------------------------------------------------------------------------
    def load_weights(self, weights: Dict[str, np.ndarray]):
        missing = [k for k in self.table if k not in weights]
        if missing:
            raise _lib.DgpError("missing tensors: %s" % ", ".join(missing[:4]))
        for k, (off, size, st) in self.table.items():
            a = np.ascontiguousarray(weights[k], dtype=np.float32)
            if a.size != size:
                raise _lib.DgpError("bad size for %s: %d != %d" % (k, a.size, size))
            self._shapes[k] = a.shape
            _lib.check(self.lib.dgp_trainer_upload(self._t, 3 if st else 0, off, a.ctypes.data_as(C.c_void_p), size))
        self.sync()
        self._fast_key = None        # new weights, new ranges: the next pass is a plain one

    def _download(self, which: int) -> Dict[str, np.ndarray]:
        out = {}
        for k, (off, size, st) in self.table.items():
            if st and which != 3 or (not st and which == 3):
                continue
            a = np.empty(size, dtype=np.float32)
            _lib.check(self.lib.dgp_trainer_download(self._t, which, off, a.ctypes.data_as(C.c_void_p), size))
            out[k] = a.reshape(self._shapes.get(k, (size,)))
        return out

    def get_weights(self) -> Dict[str, np.ndarray]:
        w = self._download(0)
        w.update(self._download(3))
        return w

    def get_grads(self) -> Dict[str, np.ndarray]:
        return self._download(1)

    def sync(self):
        _lib.check(self.lib.dgp_trainer_sync_weights(self._t, _stream(self.device)), "dgp_trainer_sync_weights")
        self._synced = True

    # ---- one optimisation step -------------------------------------------------------------------
    def workspace(self, nt: int) -> torch.Tensor:
        if self._ws is None or nt > self._ws_nt:
            n = C.c_size_t()
            _lib.check(self.lib.dgp_trainer_workspace_bytes(self._t, nt, C.byref(n)))
            if self._ws is None or self._ws.numel() < n.value:
                self._ws = torch.empty(n.value, dtype=torch.uint8, device=self.device)
            self._ws_nt = nt
        return self._ws

    def set_input_size(self, in_h: int, in_w: int):
        self.net.set_input_size(in_h, in_w)
        self._ws_nt = 0
        self._fast_key = None

    def _forward(self, frames: torch.Tensor):
        nt = frames.shape[0]
        if frames.device.type != "cuda" or not frames.is_contiguous():
            raise _lib.DgpError("frames must be a contiguous device tensor")
        if tuple(frames.shape[1:]) != (self.net.in_h, self.net.in_w, 3) or frames.dtype != torch.uint8:
            raise _lib.DgpError("frames must be uint8 [nt,%d,%d,3], got %s %s" % (self.net.in_h, self.net.in_w,
                                                                                  tuple(frames.shape), frames.dtype))
        wsb = self.workspace(nt)
        sc, lr = C.c_void_p(), C.c_void_p()
        _lib.check(self.lib.dgp_train_forward(self._t, _ptr(frames), nt, _ptr(wsb), wsb.numel(), C.byref(sc), C.byref(lr),
                                              _stream(self.device)), "dgp_train_forward")
        nj, oh, ow = self.net.nj, self.net.out_h, self.net.out_w
        return wsb, _view(sc.value, (nt, oh, ow, nj), self.device), _view(lr.value, (nt, oh, ow, 2 * nj), self.device)

    # ---- fast pass (csrc/dgp_train.hip, dgp_trainer_fast_mode): once a pass of the same shapes has left its ranges behind, the
    # retained activations of blocks 2-4 are kept as fp16 high / low tensors with PREDICTED scales.  A step whose tensors left the
    # predicted ranges reports it (one flag, read at the step's own synchronisation) and is repeated here as a plain pass before
    # anything uses its gradients -- callers never see an invalid step.
    def _fast_begin(self, nt: int) -> bool:
        key = (nt, self.net.in_h, self.net.in_w)
        # (opt-in: at 11 frames the forward convs of blocks 2-4 are grids of 52-208 tiles, paced by one tile's latency rather than by
        #  operand traffic, and the fast pass measured 12.95-13.09 against 13.01-13.16 ms per step -- EXPERIMENTS.md section 6a')
        fast = getattr(self, "_fast_key", None) == key and (
            self.tier == "f16" or                                # the 16-bit tier: every pass after the first of a shape
            (bool(self.lib.dgp_tuning_build()) and os.environ.get("DGP_TRAIN_H2", "0") == "1"))      # (H2 fast pass: an opt-in of -DDGP_TUNING builds, the only ones that read the variable)
        _lib.check(self.lib.dgp_trainer_fast_mode(self._t, 1 if fast else 0), "dgp_trainer_fast_mode")
        return fast                                  # (a REQUEST: fast_passes counts what the device reports it ran -- _fast_end / step)

    def _fast_end(self, nt: int, fast: bool) -> bool:
        """True: the pass is valid.  False: it was a fast pass that failed -- run it again (the next _fast_begin is a plain pass)."""
        if fast:
            was, failed = C.c_int32(), C.c_int32()
            _lib.check(self.lib.dgp_trainer_fast_status(self._t, C.byref(was), C.byref(failed)), "dgp_trainer_fast_status")
            if was.value:                            # the library may still have run a plain pass (no LDS-DMA weight gradients, missing ranges ...)
                self.fast_passes += 1
            if failed.value:
                self._fast_key = None
                self.fast_redos += 1
                return False
        self._fast_key = (nt, self.net.in_h, self.net.in_w)
        return True

    def forward_backward_dlc(self, *args, **kwargs):
        while True:
            fast = self._fast_begin(args[0].shape[0])
            out = self._forward_backward_dlc(*args, **kwargs)
            if self._fast_end(args[0].shape[0], fast):
                return out

    def _forward_backward_dlc(self, frames: torch.Tensor, part_score_targets, locref_targets, locref_mask,
                             part_score_weights=None, locref_loss_weight: float = 0.05, locref_huber_loss: bool = True,
                             location_refinement: bool = True):
        """One DLC step-0 loss + gradients (pose_net.train, pose_net.py:159-190): frames uint8 [nt,H,W,3] on device,
        targets fp32 device tensors of the scoremap size.  Returns {part_loss, locref_loss, total_loss}."""
        frames = frames.contiguous()
        wsb, pred, loc = self._forward(frames)
        nt = frames.shape[0]
        f32 = lambda t: None if t is None else torch.as_tensor(t, dtype=torch.float32, device=self.device).contiguous()
        pt, pw, lt, lm = f32(part_score_targets), f32(part_score_weights), f32(locref_targets), f32(locref_mask)
        if tuple(pt.shape) != tuple(pred.shape) or (location_refinement and tuple(lt.shape) != tuple(loc.shape)):
            raise _lib.DgpError("target maps %s do not match the scoremap %s" % (tuple(pt.shape), tuple(pred.shape)))
        dpred, dloc = torch.empty_like(pred), torch.zeros_like(loc)
        losses = torch.empty(4, dtype=torch.float32, device=self.device)
        scratch = torch.empty(4, dtype=torch.float64, device=self.device)
        st = _stream(self.device)
        _lib.check(self.lib.dgp_dlc_loss_fwd_bwd(_ptr(pred), _ptr(loc) if location_refinement else None, _ptr(pt),
                                                 _ptr(pw) if pw is not None else None, _ptr(lt) if location_refinement else None,
                                                 _ptr(lm) if location_refinement else None, nt, pred.shape[1], pred.shape[2],
                                                 self.net.nj, float(locref_loss_weight), int(locref_huber_loss), _ptr(dpred),
                                                 _ptr(dloc), _ptr(losses), _ptr(scratch), 32, st), "dgp_dlc_loss_fwd_bwd")
        _lib.check(self.lib.dgp_train_backward(self._t, nt, _ptr(wsb), wsb.numel(), _ptr(dpred), _ptr(dloc), st),
                   "dgp_train_backward")
        l = losses.cpu().numpy()
        return {"part_loss": float(l[0]), "locref_loss": float(l[1]), "total_loss": float(l[2])}

    def forward_backward(self, frames: torch.Tensor, *args, **kwargs):
        """frames uint8 [nt,H,W,3] on device.  Fills the gradient buffer; returns the loss dict."""
        while True:
            fast = self._fast_begin(frames.shape[0])
            out = self._forward_backward(frames, *args, **kwargs)
            if self._fast_end(frames.shape[0], fast):
                return out

    def _forward_backward(self, frames: torch.Tensor, batch: dict, hyper: DGPHyper, S0, ws, ws_max, n_frames_total,
                          n_visible_frames_total, labeled_only: bool = False, readback: bool = True):
        from .loss import dgp_loss_launch, dgp_loss_prepare, losses_to_dict
        frames = frames.contiguous()
        nt = frames.shape[0]
        st = _stream(self.device)
        nj = self.net.nj
        if labeled_only:          # fit_dgp_labeledonly: total_loss_visible, no hidden / clique terms (fitdgp.py:416)
            batch = dict(batch, hidden_marker=np.empty(0, dtype=np.int32))
            S0 = np.zeros((0, nj))
            ws = ws_max = np.zeros(0)
        # host-side order: the forward is enqueued first (it needs only the frames: 0.5 ms of host work for 2 - 4 ms of GPU work at 11
        # frames); while it runs the loss inputs are validated, packed into one pinned buffer and uploaded with ONE asynchronous copy on
        # their own stream (0.2 ms of host work).  Then the loss kernels and the whole backward pass are enqueued back to back; nothing is
        # read back here
        main = torch.cuda.current_stream(self.device)
        if getattr(self, "_upload_stream", None) is None:
            self._upload_stream = torch.cuda.Stream(device=self.device)

        def upload():
            with torch.cuda.stream(self._upload_stream):    # the uploads run beside the forward instead of queueing behind it
                li_ = dgp_loss_prepare(nt, self.net.out_h, self.net.out_w, nj, batch, hyper, S0, ws, ws_max, n_frames_total,
                                       n_visible_frames_total, self.device)
                ev = torch.cuda.Event()
                ev.record(self._upload_stream)
            return li_, ev
        wsb, pred, loc = self._forward(frames)          # checks dtype / shape against the net's current input size
        li, uploaded = upload()
        main.wait_event(uploaded)
        for name in li.__slots__:
            t = getattr(li, name, None)
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(main)
        losses, dpred, dloc, mu = dgp_loss_launch(li, pred, loc)
        _lib.check(self.lib.dgp_train_backward(self._t, nt, _ptr(wsb), wsb.numel(), _ptr(dpred), _ptr(dloc), st),
                   "dgp_train_backward")
        return losses_to_dict(losses) if readback else losses

    def grads_tensor(self) -> torch.Tensor:
        """The flat fp32 gradient buffer of every trainable tensor as a device tensor (no copy)."""
        ptr = self.lib.dgp_trainer_buffer(self._t, 1)
        return _view(ptr, (self.n_trainable,), self.device)

    def grad_groups(self):
        """[(lo, hi)] float ranges of the flat gradient buffer in the order the last backward pass completed them (dgp_trainer_grad_groups)"""
        n = C.c_int32()
        lo, hi = (C.c_int64 * 8)(), (C.c_int64 * 8)()
        _lib.check(self.lib.dgp_trainer_grad_groups(self._t, 8, C.byref(n), lo, hi), "dgp_trainer_grad_groups")
        return [(int(lo[k]), int(hi[k])) for k in range(n.value)]

    def allreduce_gradients(self, group=None, overlap: bool = True):
        """Data-parallel step: average the gradients over the ranks (RCCL) before apply_gradients.

        overlap (default): NO host synchronisation and one all-reduce per gradient GROUP, issued on a communication stream that waits for
        the event the backward pass recorded behind that group's finalisation -- the heads and block4 (60 % of the parameters) are on the
        wire while the pass is still computing blocks 3 .. 1; the caller's stream then waits for the collectives.  The result is the same
        sum (one all-reduce per contiguous range of the same buffer); overlap=False: one synchronisation, then 64-MB buckets (round 5)."""
        from .dist import average_gradients
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        groups = self.grad_groups() if overlap else []
        if not groups:
            torch.cuda.current_stream(self.device).synchronize()       # backward kernels ran on this stream via the C-ABI
            average_gradients(self.grads_tensor(), group)
            return
        flat = self.grads_tensor()
        world = dist.get_world_size(group)
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=self.device)
        cs = self._comm_stream
        handles = []
        with torch.cuda.stream(cs):
            for k, (lo, hi) in enumerate(groups):
                _lib.check(self.lib.dgp_trainer_grad_group_wait(self._t, k, C.c_void_p(cs.cuda_stream)), "dgp_trainer_grad_group_wait")
                handles.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True))
            if self.tier == "f16":
                # a 16-bit pass that left its predicted ranges on ONE rank poisons the sum on every rank: all ranks must skip the update
                # (the optimiser's kernel reads the flag) and repeat the step on the parity path together
                flag = _view(self.lib.dgp_trainer_buffer(self._t, 4), (1,), self.device, "<i4")
                handles.append(dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group, async_op=True))
            for h in handles:
                h.wait()                                               # (the communication stream waits for the collective)
            flat.mul_(1.0 / world)
        torch.cuda.current_stream(self.device).wait_stream(cs)         # the optimiser's kernels run behind the averaged gradients

    def apply_gradients(self, lr: float, momentum: float = 0.9, clip_norm: float = 10.0) -> float:
        g = C.c_float()
        _lib.check(self.lib.dgp_sgd_momentum_clip(self._t, lr, momentum, clip_norm, C.byref(g), _stream(self.device)),
                   "dgp_sgd_momentum_clip")
        self.sync()
        return g.value

    def step(self, frames, batch, hyper: DGPHyper, S0, ws, ws_max, n_frames_total, n_visible_frames_total,
             labeled_only: bool = False):
        """One optimisation step = the reference's sess.run([loss, train_op]) (DGP/models/fitdgp.py:818).  Single process: the whole step --
        forward, loss, backward, clip + momentum, the re-packing of the weights -- is enqueued without a read-back and ONE synchronisation
        at its end fetches losses, gradient norm and the pass status (dgp_trainer_step_status).  A 16-bit pass whose tensors left their
        predicted ranges has skipped its update on the device and is repeated here on the parity path."""
        import torch.distributed as dist
        dp = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        from .loss import LOSS_NAMES
        nt = frames.shape[0]
        while True:
            fast = self._fast_begin(nt)
            dev_losses = self._forward_backward(frames, batch, hyper, S0, ws, ws_max, n_frames_total, n_visible_frames_total, labeled_only,
                                                readback=False)
            if dp:
                # data-parallel: the mean gradient over the ranks (RCCL), group by group while the backward pass is still running; the
                # optimiser's kernels are enqueued behind it -- still ONE host synchronisation per step
                self.allreduce_gradients()
            _lib.check(self.lib.dgp_sgd_momentum_clip(self._t, hyper.lr, hyper.momentum, hyper.clip_norm, None, _stream(self.device)),
                       "dgp_sgd_momentum_clip")
            self.sync()                              # master -> panels / cells for the next forward (same weights again after a skipped update)
            g, was, failed = C.c_float(), C.c_int32(), C.c_int32()
            lv = (C.c_float * 8)()
            _lib.check(self.lib.dgp_trainer_step_status(self._t, _ptr(dev_losses), len(LOSS_NAMES), lv, C.byref(g), C.byref(was), C.byref(failed),
                                                        _stream(self.device)), "dgp_trainer_step_status")
            if fast and was.value:
                self.fast_passes += 1                # counted from the pass status the DEVICE wrote, not from the request
            if fast and failed.value:                # the momentum kernel saw the flag and left parameters and momentum alone
                self._fast_key = None
                self.fast_redos += 1
                continue
            self._fast_key = (nt, self.net.in_h, self.net.in_w)
            losses = {k: float(lv[i]) for i, k in enumerate(LOSS_NAMES)}
            losses["grad_norm"] = g.value
            return losses


def _view(ptr: int, shape, device, typestr: str = "<f4") -> torch.Tensor:
    """torch view of a device pointer inside the (torch-owned) workspace / the library's flat buffers (fp32, or int32 with typestr "<i4")."""
    # a tensor over foreign device memory through the __cuda_array_interface__ protocol
    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}
    return torch.as_tensor(h, device=device)
