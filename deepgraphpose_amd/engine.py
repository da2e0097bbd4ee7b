"""Torch-facing wrapper over the C-ABI (include/dgp_hip.h).

torch is plumbing here: it owns device memory and streams; every computation on the hot
path is a HIP kernel in libdgp_hip.so reached through ctypes.  Nothing in this module
falls back to torch ops or to the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .arch import MEAN_PIXEL, BN_EPS


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _need_cuda(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise _lib.DgpError("%s must be a device (cuda/HIP) tensor" % name)
    if t.dtype != dtype:
        raise _lib.DgpError("%s must have dtype %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise _lib.DgpError("%s must be contiguous" % name)


class DGPNet:
    """ResNet-v1 backbone + DGP heads on one MI355X.

    Replaces the TF graph built by setup_dgp_eval_graph (DGP/models/eval.py:147-214):
    `forward` is sess.run(scmap), `infer` is the whole per-frame body of estimate_pose's loop
    (eval.py:306-345) for a batch of frames.
    """

    def __init__(self, depth: int = 50, num_joints: int = 4, in_h: int = 480, in_w: int = 640,
                 max_batch: int = 32, with_locref: bool = False, device: int = 0,
                 mean_pixel=MEAN_PIXEL, bn_eps: float = BN_EPS, tier: Optional[str] = None):
        """tier: None = the library's default (the parity tier, or the 16-bit tier under DGP_CONV_MODE=f16); "parity" / "f32x": fp32-class
        arithmetic on H2 cells (the 1e-3 px / bit-exact-index gate); "f16": the 16-bit tier -- 2-byte H1 activation cells, one MFMA per
        product (include/dgp_hip.h, dgp_net_set_tier): a reported tier with measured error, not a parity claim."""
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.DgpError("no HIP device visible: DGPNet needs an MI355X (there is no CPU fallback)")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        d = _lib.DgpNetDesc(depth, num_joints, in_h, in_w, max_batch, int(with_locref),
                            (C.c_float * 3)(*mean_pixel), bn_eps)
        h = C.c_void_p()
        _lib.check(self.lib.dgp_net_create(C.byref(d), C.byref(h)), "dgp_net_create")
        self._h = h
        self.depth, self.nj, self.in_h, self.in_w = depth, num_joints, in_h, in_w
        self.max_batch, self.with_locref = max_batch, with_locref
        oh, ow, fh, fw = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self.lib.dgp_net_output_dims(h, C.byref(oh), C.byref(ow), C.byref(fh), C.byref(fw)))
        self.out_h, self.out_w, self.feat_h, self.feat_w = oh.value, ow.value, fh.value, fw.value
        self._ws: Optional[torch.Tensor] = None
        self._ws_batch = 0
        # bookkeeping of the H2 activation scales for callers that keep several engines in step (DGPPipeline): widen_count = how often
        # the headroom was raised since load_weights, scale_epoch = bumped by everything that invalidates the calibrated scales
        self.widen_count = 0
        self.scale_epoch = 0
        if tier is not None:
            self.set_tier(tier)

    TIERS = {"parity": 0, "f32x": 0, "f16": 1}

    def set_tier(self, tier):
        t = self.TIERS[tier] if isinstance(tier, str) else int(tier)
        _lib.check(self.lib.dgp_net_set_tier(self._h, t), "dgp_net_set_tier")
        self.scale_epoch += 1                   # the next forward calibrates again

    @property
    def tier(self) -> str:
        return "f16" if self.lib.dgp_net_get_tier(self._h) == 1 else "parity"

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.dgp_net_destroy(h)
            self._h = None

    # -- weights ---------------------------------------------------------------------
    def load_weights(self, weights: Dict[str, np.ndarray]):
        """weights: TF variable name -> fp32 array in TF layout (see synthetic.make_weights)."""
        keep = []
        views = (_lib.DgpTensorView * len(weights))()
        for i, (k, v) in enumerate(weights.items()):
            a = np.ascontiguousarray(v, dtype=np.float32)
            keep.append(a)
            views[i].name = k.encode()
            views[i].data = a.ctypes.data_as(C.POINTER(C.c_float))
            views[i].ndim = a.ndim
            for j in range(4):
                views[i].shape[j] = a.shape[j] if j < a.ndim else 1
        _lib.check(self.lib.dgp_net_load_weights(self._h, views, len(weights)), "dgp_net_load_weights")
        self.widen_count = 0                    # (the library resets the headroom with the weights)
        self.scale_epoch += 1

    def set_input_size(self, in_h: int, in_w: int):
        """Re-plan the geometry for another frame size; weights stay (DLC's step-0 loader changes the size every
        iteration, pose_defaultdataset.py:131-196)."""
        if (in_h, in_w) == (self.in_h, self.in_w):
            return
        _lib.check(self.lib.dgp_net_set_input_size(self._h, in_h, in_w), "dgp_net_set_input_size")
        self.in_h, self.in_w = in_h, in_w
        oh, ow, fh, fw = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self.lib.dgp_net_output_dims(self._h, C.byref(oh), C.byref(ow), C.byref(fh), C.byref(fw)))
        self.out_h, self.out_w, self.feat_h, self.feat_w = oh.value, ow.value, fh.value, fw.value
        self._ws_batch = 0                      # workspace is re-checked against the new geometry

    # -- workspace -------------------------------------------------------------------
    def workspace(self, batch: int) -> torch.Tensor:
        if self._ws is None or batch > self._ws_batch:
            n = C.c_size_t()
            _lib.check(self.lib.dgp_net_workspace_bytes(self._h, batch, C.byref(n)), "dgp_net_workspace_bytes")
            if self._ws is None or self._ws.numel() < n.value:
                self._ws = torch.empty(n.value, dtype=torch.uint8, device=self.device)
            self._ws_batch = batch
        return self._ws

    def stats(self, batch: int) -> Tuple[int, float]:
        nl, fl = C.c_int32(), C.c_double()
        _lib.check(self.lib.dgp_net_stats(self._h, batch, C.byref(nl), C.byref(fl)))
        return nl.value, fl.value

    # -- measurement -----------------------------------------------------------------
    def profile_begin(self, max_steps: int):
        _lib.check(self.lib.dgp_net_profile_begin(self._h, max_steps), "dgp_net_profile_begin")

    def profile_end(self):
        """-> list of (name, algorithmic flops, avg ms) per launch of one step."""
        ns, nl = C.c_int32(), C.c_int32()
        _lib.check(self.lib.dgp_net_profile_end(self._h, C.byref(ns), C.byref(nl)), "dgp_net_profile_end")
        out = []
        buf = C.create_string_buffer(256)
        for i in range(nl.value):
            fl, ms = C.c_double(), C.c_double()
            _lib.check(self.lib.dgp_net_profile_launch(self._h, i, buf, 256, C.byref(fl), C.byref(ms)))
            if buf.value:
                out.append((buf.value.decode(), fl.value, ms.value))
        return ns.value, out

    # -- compute ---------------------------------------------------------------------
    def forward(self, frames: torch.Tensor, want_locref: bool = False, want_features: bool = False, check_range: bool = True):
        """frames uint8 [B,H,W,3] on device -> scmap [B,out_h,out_w,nj] (and locref / features).  check_range: as in infer()."""
        _need_cuda(frames, torch.uint8, "frames")
        B = frames.shape[0]
        if tuple(frames.shape[1:]) != (self.in_h, self.in_w, 3):
            raise _lib.DgpError("frames must be [B,%d,%d,3], got %s" % (self.in_h, self.in_w, tuple(frames.shape)))
        scmap = torch.empty((B, self.out_h, self.out_w, self.nj), dtype=torch.float32, device=self.device)
        locref = torch.empty((B, self.out_h, self.out_w, 2 * self.nj), dtype=torch.float32,
                             device=self.device) if want_locref else None
        feats = torch.empty((B, self.feat_h, self.feat_w, 2048), dtype=torch.float32,
                            device=self.device) if want_features else None
        ws = self.workspace(B) if B else None
        for attempt in range(5 if B else 0):           # (an empty batch -- e.g. the empty shard of a short video -- gives empty outputs, like sess.run)
            _lib.check(self.lib.dgp_forward(self._h, _ptr(frames), B, _ptr(ws), ws.numel(), _ptr(scmap), _ptr(locref),
                                            _ptr(feats), _stream(self.device)), "dgp_forward")
            if not check_range or not self.range_status()[0]:
                break
        else:
            if B:
                raise _lib.DgpError("dgp_forward: activation ranges still overflow after 4 re-calibrations (non-finite weights or input?)")
        out = [scmap]
        if want_locref:
            out.append(locref)
        if want_features:
            out.append(feats)
        return out[0] if len(out) == 1 else tuple(out)

    def infer(self, frames: torch.Tensor, gamma: float = 1.0, gauss_len: int = 1, out=None,
              scmap_out: Optional[torch.Tensor] = None, check_range: bool = True):
        """Fused frames -> (mu [B,nj,2] (row,col), conf [B,nj], idx [B,nj,2] int32).

        check_range (default): synchronise and read the H2 range flag after the forward; a batch that outgrew the calibrated
        activation scales is re-run (the engine re-calibrates on it, up to 4 times, then DgpError).  Streaming callers pass
        check_range=False and poll range_status() themselves at a point where they can re-run (eval.estimate_pose, bench.py)."""
        _need_cuda(frames, torch.uint8, "frames")
        B = frames.shape[0]
        if tuple(frames.shape[1:]) != (self.in_h, self.in_w, 3):
            raise _lib.DgpError("frames must be [B,%d,%d,3], got %s" % (self.in_h, self.in_w, tuple(frames.shape)))
        if out is None:
            mu = torch.empty((B, self.nj, 2), dtype=torch.float32, device=self.device)
            conf = torch.empty((B, self.nj), dtype=torch.float32, device=self.device)
            idx = torch.empty((B, self.nj, 2), dtype=torch.int32, device=self.device)
        else:
            mu, conf, idx = out
        if B == 0:                                      # an empty batch gives empty outputs, like sess.run
            return mu, conf, idx
        ws = self.workspace(B)
        for attempt in range(5):
            _lib.check(self.lib.dgp_infer(self._h, _ptr(frames), B, _ptr(ws), ws.numel(), float(gamma), int(gauss_len),
                                          _ptr(mu), _ptr(conf), _ptr(idx), _ptr(scmap_out), _stream(self.device)),
                       "dgp_infer")
            if not check_range or not self.range_status()[0]:
                return mu, conf, idx
        raise _lib.DgpError("dgp_infer: activation ranges still overflow after 4 re-calibrations (non-finite weights or input?)")


    def range_status(self) -> Tuple[bool, int]:
        """(overflow, calibrations).  Synchronises the stream.  overflow: a forward since the last call outgrew the calibrated scale of
        one of its H2 activation tensors (include/dgp_hip.h, "H2"): its results are invalid, the flag is cleared and the next
        forward re-calibrates on its own batch -- re-run what was computed since the last clean status."""
        ov, nc = C.c_int32(), C.c_int32()
        _lib.check(self.lib.dgp_net_range_status(self._h, C.byref(ov), C.byref(nc), _stream(self.device)), "dgp_net_range_status")
        if ov.value:                       # the library widened this engine's headroom and will re-calibrate
            self.widen_count += 1
            self.scale_epoch += 1
        return bool(ov.value), nc.value

    def recalibrate(self):
        """Force a calibration pass of the activation scales on the next forward."""
        _lib.check(self.lib.dgp_net_recalibrate(self._h), "dgp_net_recalibrate")
        self.scale_epoch += 1

    def reset_scales(self):
        """Default headroom again and a calibration pass on the next forward: the state right after load_weights (an engine kept between
        videos starts every video as a fresh one would)."""
        _lib.check(self.lib.dgp_net_reset_scales(self._h), "dgp_net_reset_scales")
        self.widen_count = 0
        self.scale_epoch += 1

    def copy_scales_from(self, other: "DGPNet"):
        """Take `other`'s calibrated activation scales (same network, weights, tier, frame size): what this engine would find by calibrating
        on the same batch itself, without the layer-by-layer pass (dgp_net_copy_scales)."""
        _lib.check(self.lib.dgp_net_copy_scales(self._h, other._h, _stream(self.device)), "dgp_net_copy_scales")
        self.widen_count = other.widen_count
        self.scale_epoch += 1

    def widen(self):
        """Re-calibrate on the next forward with 3 more bits of headroom -- what range_status() does on an overflow -- for a rank
        that follows another rank's overflow in a sharded run."""
        _lib.check(self.lib.dgp_net_widen(self._h), "dgp_net_widen")
        self.widen_count += 1
        self.scale_epoch += 1

    def infer_packed(self, frames: torch.Tensor, traj: torch.Tensor, gamma: float = 1.0, gauss_len: int = 1,
                     scmap_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Fused frames -> traj [B,nj,5] fp32 lanes (row, col, likelihood, iy, ix; indices as int32 bit patterns): the record
        layout of the per-video trajectory (dist.unpack_keypoints splits it).  `traj` is a caller-owned contiguous slice of the
        trajectory buffer, written in place by the soft-argmax kernel."""
        _need_cuda(frames, torch.uint8, "frames")
        _need_cuda(traj, torch.float32, "traj")
        B = frames.shape[0]
        if tuple(frames.shape[1:]) != (self.in_h, self.in_w, 3):
            raise _lib.DgpError("frames must be [B,%d,%d,3], got %s" % (self.in_h, self.in_w, tuple(frames.shape)))
        if tuple(traj.shape) != (B, self.nj, 5):
            raise _lib.DgpError("traj must be [%d,%d,5], got %s" % (B, self.nj, tuple(traj.shape)))
        ws = self.workspace(B)
        _lib.check(self.lib.dgp_infer_packed(self._h, _ptr(frames), B, _ptr(ws), ws.numel(), float(gamma), int(gauss_len),
                                             _ptr(traj), _ptr(scmap_out), _stream(self.device)), "dgp_infer_packed")
        return traj


class DGPPipeline:
    """Two engines on two HIP streams, batches dealt to them in turn.

    A forward is a chain of ~50 dependent launches; most layers' grids do not fill a whole number of rounds of the 512 resident
    workgroups, so the tail of every layer runs on a partly idle chip.  With a second, independent batch in flight on another
    stream the hardware fills those tails with the other batch's workgroups: +5 % frames/s on ResNet-50 640x480 batch 32
    (scripts/two_stream_probe.py: 4279 -> 4492 over 300 steps; a third stream adds nothing).  Every engine keeps its own repacked
    weights and workspace (one forward at a time per dgp_net, include/dgp_hip.h); all engines calibrate their activation scales on
    the SAME batch (`calibrate`), so which engine a batch lands on does not change a bit of its result.

    submit() enqueues one batch and returns at once; join() makes the caller's stream wait for everything submitted."""

    def __init__(self, depth: int = 50, num_joints: int = 4, in_h: int = 480, in_w: int = 640, max_batch: int = 32,
                 with_locref: bool = False, device: int = 0, n_streams: int = 2, mean_pixel=MEAN_PIXEL,
                 first: Optional[DGPNet] = None, tier: Optional[str] = None):
        """first: an existing engine of the same configuration to adopt as engine 0 (its weights stay loaded); tier: as DGPNet's."""
        if n_streams < 1:
            raise _lib.DgpError("n_streams must be >= 1")
        self.nets = [first] if first is not None else []
        while len(self.nets) < n_streams:
            self.nets.append(DGPNet(depth, num_joints, in_h, in_w, max_batch, with_locref, device, mean_pixel, tier=tier))
        if tier is not None and first is not None:
            want = "f16" if DGPNet.TIERS[tier] == 1 else "parity"      # compare with the tier that was ASKED for (with n_streams == 1 there is no second engine)
            if first.tier != want:
                first.set_tier(tier)
        self.device = self.nets[0].device
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(n_streams)]
        self.nj, self.max_batch = num_joints, max_batch
        self._next = 0
        self._calibrated = False
        self._epochs = [n.scale_epoch for n in self.nets]
        n0 = self.nets[0]
        self.in_h, self.in_w, self.out_h, self.out_w = n0.in_h, n0.in_w, n0.out_h, n0.out_w

    def load_weights(self, weights: Dict[str, np.ndarray]):
        for n in self.nets:
            n.load_weights(weights)
        self._calibrated = False

    def set_input_size(self, in_h: int, in_w: int):
        for n in self.nets:
            n.set_input_size(in_h, in_w)
        n0 = self.nets[0]
        self.in_h, self.in_w, self.out_h, self.out_w = n0.in_h, n0.in_w, n0.out_h, n0.out_w

    def calibrate(self, frames: torch.Tensor, gamma: float = 1.0, gauss_len: int = 1):
        """Engine 0 runs `frames` (it calibrates on them after load_weights / recalibrate / an overflow) and the other engines take its
        scales (dgp_net_copy_scales: the calibration is deterministic, so that is what each would have found on the same batch -- round 6,
        one layer-by-layer pass per video instead of one per engine); synchronises."""
        scratch = torch.empty((frames.shape[0], self.nj, 5), dtype=torch.float32, device=self.device)
        cur = torch.cuda.current_stream(self.device)
        n0, st0 = self.nets[0], self.streams[0]
        st0.wait_stream(cur)
        with torch.cuda.stream(st0):
            n0.infer_packed(frames, scratch, gamma, gauss_len)
        st0.synchronize()
        for n, st in zip(self.nets[1:], self.streams[1:]):
            with torch.cuda.stream(st):
                n.copy_scales_from(n0)
        self._calibrated = True
        self._epochs = [n.scale_epoch for n in self.nets]

    def submit(self, frames: torch.Tensor, traj: torch.Tensor, gamma: float = 1.0, gauss_len: int = 1) -> "torch.cuda.Event":
        """infer_packed(frames -> traj) on the next engine's stream, ordered after the work already on the caller's current stream.
        Returns the event recorded behind it (frames / traj may be reused once it has completed).  The first batch after
        load_weights / recalibrate / an overflow goes through EVERY engine first (calibrate), so all engines share its scales."""
        if self._calibrated and [n.scale_epoch for n in self.nets] != self._epochs:
            self._resync()                  # an engine was re-calibrated behind the pipeline's back (EvalSession shares engine 0)
        if not self._calibrated:
            self.calibrate(frames, gamma, gauss_len)
        i = self._next
        self._next = (i + 1) % len(self.nets)
        st = self.streams[i]
        st.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(st):
            self.nets[i].infer_packed(frames, traj, gamma, gauss_len)
            ev = torch.cuda.Event()
            ev.record(st)
        return ev

    def _resync(self):
        """Engines whose scales were touched individually (a synchronous forward on a shared engine overflowed and re-calibrated on its
        own batch): give every engine the widest headroom any of them has and calibrate all of them again on ONE batch -- the next
        one submitted -- so that which engine a batch lands on does not change a bit of its result."""
        target = max(n.widen_count for n in self.nets)
        for n in self.nets:
            while n.widen_count < target:
                n.widen()
            n.recalibrate()
        self._calibrated = False

    def join(self):
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)

    def range_status(self) -> Tuple[bool, int]:
        """(any engine overflowed, calibrations of the first engine); an overflow on one engine widens ALL of them, so that they
        keep identical scales after the re-calibration (calibrate() again before re-running).  Waits for every engine's stream first:
        the flag is written by the range check at the end of each forward, on the engine's own (non-blocking) stream."""
        for st in self.streams:
            st.synchronize()
        res = [n.range_status() for n in self.nets]
        ov = any(r[0] for r in res)
        if ov:
            for n, r in zip(self.nets, res):
                if not r[0]:
                    n.widen()
            self._calibrated = False
        return ov, res[0][1]

    def recalibrate(self):
        for n in self.nets:
            n.recalibrate()
        self._calibrated = False

    def reset_scales(self):
        for n in self.nets:
            n.reset_scales()
        self._calibrated = False

    def widen(self):
        for n in self.nets:
            n.widen()
        self._calibrated = False


# ---------------------------------------------------------------------------------------
# stand-alone operators
# ---------------------------------------------------------------------------------------
def soft_argmax(scmap: torch.Tensor, gamma: float = 1.0, gauss_len: int = 2, want_pmap: bool = False):
    """HIP argmax_2d_from_cm (+ likelihood window).  scmap fp32 [B,H,W,C] on device."""
    lib = _lib.load()
    _need_cuda(scmap, torch.float32, "scmap")
    if scmap.dim() != 4:
        raise _lib.DgpError("scmap must be rank 4 [B,H,W,C]")      # fitdgp_util.py:357 assert rank == 4
    B, H, W, Cn = scmap.shape
    dev = scmap.device
    mu = torch.empty((B, Cn, 2), dtype=torch.float32, device=dev)
    conf = torch.empty((B, Cn), dtype=torch.float32, device=dev)
    idx = torch.empty((B, Cn, 2), dtype=torch.int32, device=dev)
    pmap = torch.empty_like(scmap) if want_pmap else None
    if B * Cn == 0 and H > 0 and W > 0:                 # no frames or no joints: empty outputs
        return (mu, conf, idx, pmap) if want_pmap else (mu, conf, idx)
    _lib.check(lib.dgp_soft_argmax(_ptr(scmap), B, H, W, Cn, float(gamma), int(gauss_len), _ptr(mu), _ptr(conf),
                                   _ptr(idx), _ptr(pmap), _stream(dev)), "dgp_soft_argmax")
    return (mu, conf, idx, pmap) if want_pmap else (mu, conf, idx)


def pmap_threshold(pmap: torch.Tensor, th: float) -> torch.Tensor:
    """argmax_2d_from_cm's `th` branch (fitdgp_util.py:377-388): pmap [B,H,W,C] (normalised blurred softmax from soft_argmax) is
    thresholded at th * max per (frame, joint) map and renormalised IN PLACE; returns mu [B,C,2] (row, col) of the new maps."""
    lib = _lib.load()
    _need_cuda(pmap, torch.float32, "pmap")
    if pmap.dim() != 4 or not pmap.is_contiguous():
        raise _lib.DgpError("pmap must be a contiguous rank-4 tensor [B,H,W,C]")
    B, H, W, Cn = pmap.shape
    mu = torch.empty((B, Cn, 2), dtype=torch.float32, device=pmap.device)
    if B * Cn == 0 and H > 0 and W > 0:
        return mu
    _lib.check(lib.dgp_pmap_threshold(_ptr(pmap), B, H, W, Cn, float(th), _ptr(mu), _stream(pmap.device)), "dgp_pmap_threshold")
    return mu


def hard_argmax(scmap: torch.Tensor, locref: Optional[torch.Tensor] = None):
    """HIP DLC arg-max: idx [B,C,2] (row,col), prob [B,C], offs [B,C,2] (dx,dy raw locref)."""
    lib = _lib.load()
    _need_cuda(scmap, torch.float32, "scmap")
    B, H, W, Cn = scmap.shape
    if locref is not None:
        _need_cuda(locref, torch.float32, "locref")
        if tuple(locref.shape) != (B, H, W, 2 * Cn):
            raise _lib.DgpError("locref must be [B,H,W,2*C]")
    dev = scmap.device
    idx = torch.empty((B, Cn, 2), dtype=torch.int32, device=dev)
    prob = torch.empty((B, Cn), dtype=torch.float32, device=dev)
    offs = torch.empty((B, Cn, 2), dtype=torch.float32, device=dev)
    if B * Cn == 0 and H > 0 and W > 0:
        return idx, prob, offs
    _lib.check(lib.dgp_hard_argmax(_ptr(scmap), _ptr(locref), B, H, W, Cn, _ptr(idx), _ptr(prob), _ptr(offs),
                                   _stream(dev)), "dgp_hard_argmax")
    return idx, prob, offs


def pack_conv_weights(w_hwio: np.ndarray) -> np.ndarray:
    lib = _lib.load()
    w = np.ascontiguousarray(w_hwio, dtype=np.float32)
    kh, kw, cin, cout = w.shape
    n = lib.dgp_packed_weight_floats(kh, kw, cin, cout)
    if n == 0:
        raise _lib.DgpError("unsupported conv weight shape %s (Cin must be 4*2^k)" % (w.shape,))
    out = np.empty(n, dtype=np.float32)
    _lib.check(lib.dgp_pack_conv_weights(w.ctypes.data_as(C.c_void_p), kh, kw, cin, cout,
                                         out.ctypes.data_as(C.c_void_p)), "dgp_pack_conv_weights")
    return out


ABSMAX_SLOTS = 256      # DGP_ABSMAX_SLOTS in include/dgp_hip.h


def conv2d(x: torch.Tensor, w_hwio: np.ndarray, stride: int = 1, rate: int = 1, pad_t: int = 0, pad_l: int = 0,
           out_hw: Optional[Tuple[int, int]] = None, scale=None, bias=None, residual: Optional[torch.Tensor] = None,
           res_stride: int = 0, relu: bool = False, ranged: bool = False, return_range: bool = False):
    """Single conv layer through the implicit-GEMM kernel (NHWC fp32).

    ranged=True measures max |x| and max |w| on the device first (dgp_tensor_absmax) and passes them on, which lets
    the fp16 high/low split kernels run (dgp_conv2d_ranged); return_range=True also returns max |y| as tracked by the
    epilogue (a device tensor of DGP_ABSMAX_SLOTS floats whose maximum is the value)."""
    lib = _lib.load()
    _need_cuda(x, torch.float32, "x")
    N, H, W, Cin = x.shape
    kh, kw, cin2, cout = w_hwio.shape
    assert cin2 == Cin
    dev = x.device
    if out_hw is None:
        keh, kew = (kh - 1) * rate + 1, (kw - 1) * rate + 1
        out_hw = ((H + 2 * pad_t - keh) // stride + 1, (W + 2 * pad_l - kew) // stride + 1)
    Ho, Wo = out_hw
    wp = torch.from_numpy(pack_conv_weights(w_hwio)).to(dev)
    sc = None if scale is None else torch.as_tensor(scale, dtype=torch.float32).contiguous().to(dev)
    bi = None if bias is None else torch.as_tensor(bias, dtype=torch.float32).contiguous().to(dev)
    y = torch.empty((N, Ho, Wo, cout), dtype=torch.float32, device=dev)
    rh, rw = (residual.shape[1], residual.shape[2]) if residual is not None else (0, 0)
    d = _lib.DgpConvDesc(N, H, W, Cin, cout, kh, kw, stride, rate, pad_t, pad_l, Ho, Wo, int(relu),
                         res_stride if residual is not None else 0, rh, rw)
    if not (ranged or return_range):
        _lib.check(lib.dgp_conv2d(C.byref(d), _ptr(x), _ptr(wp), _ptr(sc), _ptr(bi), _ptr(residual), _ptr(y),
                                  _stream(dev)), "dgp_conv2d")
        return y
    rng = torch.zeros((3, ABSMAX_SLOTS), dtype=torch.float32, device=dev)
    if ranged:
        _lib.check(lib.dgp_tensor_absmax(_ptr(x), x.numel(), _ptr(rng[0]), _stream(dev)), "dgp_tensor_absmax")
        _lib.check(lib.dgp_tensor_absmax(_ptr(wp), wp.numel(), _ptr(rng[1]), _stream(dev)), "dgp_tensor_absmax")
    _lib.check(lib.dgp_conv2d_ranged(C.byref(d), _ptr(x), _ptr(wp), _ptr(sc), _ptr(bi), _ptr(residual), _ptr(y),
                                     _ptr(rng[0]) if ranged else None, _ptr(rng[1]) if ranged else None, _ptr(rng[2]),
                                     _stream(dev)), "dgp_conv2d_ranged")
    return (y, rng[2]) if return_range else y


def _conv_desc(x_shape, w_shape, stride, rate, pad_t, pad_l, out_hw):
    N, H, W, Cin = x_shape
    kh, kw, _, cout = w_shape
    Ho, Wo = out_hw
    return _lib.DgpConvDesc(N, H, W, Cin, cout, kh, kw, stride, rate, pad_t, pad_l, Ho, Wo, 0, 0, 0, 0)


def conv2d_wgrad(x: torch.Tensor, dy: torch.Tensor, ksize: int, stride: int = 1, rate: int = 1, pad_t: int = 0, pad_l: int = 0,
                 ranged: bool = True):
    """Weight gradient of one conv layer through the trainer's kernels: x [N,H,W,Cin], dy [N,Ho,Wo,Cout] ->
    (dW [k,k,Cin,Cout] HWIO, colsum [Cout]).  ranged: measure both operand ranges first (fp16-split kernel on 128 x 128 tiles)."""
    lib = _lib.load()
    _need_cuda(x, torch.float32, "x")
    _need_cuda(dy, torch.float32, "dy")
    dev = x.device
    N, H, W, Cin = x.shape
    _, Ho, Wo, Cout = dy.shape
    d = _conv_desc(x.shape, (ksize, ksize, Cin, Cout), stride, rate, pad_t, pad_l, (Ho, Wo))
    dw = torch.empty((ksize, ksize, Cin, Cout), dtype=torch.float32, device=dev)
    cs = torch.empty(2 * Cout, dtype=torch.float32, device=dev)
    rng = torch.zeros((2, ABSMAX_SLOTS), dtype=torch.float32, device=dev)
    if ranged:
        _lib.check(lib.dgp_tensor_absmax(_ptr(x), x.numel(), _ptr(rng[0]), _stream(dev)), "dgp_tensor_absmax")
        _lib.check(lib.dgp_tensor_absmax(_ptr(dy), dy.numel(), _ptr(rng[1]), _stream(dev)), "dgp_tensor_absmax")
    _lib.check(lib.dgp_conv2d_wgrad(C.byref(d), _ptr(x), _ptr(dy), _ptr(rng[0]) if ranged else None,
                                    _ptr(rng[1]) if ranged else None, _ptr(dw), _ptr(cs), _stream(dev)), "dgp_conv2d_wgrad")
    return dw, cs[:Cout]


def conv2d_wgrad_shadow(x: torch.Tensor, dy: torch.Tensor, ksize: int, stride: int = 1, rate: int = 1, pad_t: int = 0, pad_l: int = 0,
                        prev_ratio: Tuple[float, float] = (1.0, 1.0)):
    """conv2d_wgrad through the training step's LDS-DMA tile: both operands are copied into fp16 high / low cells with the scales a
    "previous step" whose maxima were prev_ratio times this step's would predict (1.0: the steady state; 2^-6 or 2^8: the prediction
    fails and the kernel computes the tile from the fp32 tensors; 0: no previous range at all, the first step)."""
    lib = _lib.load()
    _need_cuda(x, torch.float32, "x")
    _need_cuda(dy, torch.float32, "dy")
    dev = x.device
    N, H, W, Cin = x.shape
    _, Ho, Wo, Cout = dy.shape
    d = _conv_desc(x.shape, (ksize, ksize, Cin, Cout), stride, rate, pad_t, pad_l, (Ho, Wo))
    dw = torch.empty((ksize, ksize, Cin, Cout), dtype=torch.float32, device=dev)
    cs = torch.empty(2 * Cout, dtype=torch.float32, device=dev)
    rng = torch.zeros((2, ABSMAX_SLOTS), dtype=torch.float32, device=dev)
    _lib.check(lib.dgp_tensor_absmax(_ptr(x), x.numel(), _ptr(rng[0]), _stream(dev)), "dgp_tensor_absmax")
    _lib.check(lib.dgp_tensor_absmax(_ptr(dy), dy.numel(), _ptr(rng[1]), _stream(dev)), "dgp_tensor_absmax")
    prev = torch.stack([rng[0] * float(prev_ratio[0]), rng[1] * float(prev_ratio[1])]).contiguous()
    scratch = torch.empty(x.numel() + dy.numel(), dtype=torch.float32, device=dev)
    _lib.check(lib.dgp_conv2d_wgrad_shadow(C.byref(d), _ptr(x), _ptr(dy), _ptr(rng[0]), _ptr(rng[1]), _ptr(prev[0]), _ptr(prev[1]),
                                           _ptr(scratch), _ptr(dw), _ptr(cs), _stream(dev)), "dgp_conv2d_wgrad_shadow")
    return dw, cs[:Cout]


def conv2d_dgrad(dy: torch.Tensor, w_hwio: torch.Tensor, x_hw: Tuple[int, int], stride: int = 1, rate: int = 1, pad_t: int = 0,
                 pad_l: int = 0, scale: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                 dx_add: Optional[torch.Tensor] = None, add_mode: int = 1, ranged: bool = True, mask_h2: bool = False) -> torch.Tensor:
    """Data gradient of one conv layer through the trainer's kernels: dy [N,Ho,Wo,Cout], w [k,k,Cin,Cout] (device) -> dx [N,H,W,Cin].
    mask_h2: hand the gate tensor over as H2 cells (the fast pass of the training step keeps its activations that way)."""
    lib = _lib.load()
    _need_cuda(dy, torch.float32, "dy")
    _need_cuda(w_hwio, torch.float32, "w_hwio")
    dev = dy.device
    N, Ho, Wo, Cout = dy.shape
    kh, kw, Cin, _ = w_hwio.shape
    H, W = x_hw
    d = _conv_desc((N, H, W, Cin), w_hwio.shape, stride, rate, pad_t, pad_l, (Ho, Wo))
    dx = torch.empty((N, H, W, Cin), dtype=torch.float32, device=dev)
    scratch = torch.empty(lib.dgp_conv2d_dgrad_scratch_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
    flags = int(bool(ranged))
    if mask_h2 and mask is not None:
        mask = f32_to_h2(mask.contiguous(), h2_exp_for(float(mask.abs().max())))
        flags |= 2
    _lib.check(lib.dgp_conv2d_dgrad(C.byref(d), _ptr(dy), _ptr(w_hwio), _ptr(scale), _ptr(mask), _ptr(dx_add), add_mode, _ptr(dx),
                                    _ptr(scratch), flags, _stream(dev)), "dgp_conv2d_dgrad")
    return dx


def h2_exp_for(absmax: float, headroom_bits: int = 4) -> int:
    """The engine's scale rule: exponent e with absmax * 2^e in [2^(14 - headroom), 2^(15 - headroom))."""
    import math
    if not (absmax > 0) or not math.isfinite(absmax):
        return 0
    return (14 - headroom_bits) - (math.frexp(absmax)[1] - 1)


def f32_to_h2(x: torch.Tensor, scale_exp: int) -> torch.Tensor:
    """fp32 [..., C] (C % 8 == 0) -> H2 cells (fp16 high / low pairs per 8 channels, x * 2^scale_exp), same shape, viewed as fp32 bits."""
    lib = _lib.load()
    _need_cuda(x, torch.float32, "x")
    out = torch.empty_like(x)
    _lib.check(lib.dgp_f32_to_h2(_ptr(x), x.numel(), int(scale_exp), _ptr(out), _stream(x.device)), "dgp_f32_to_h2")
    return out


def h2_to_f32(x: torch.Tensor, scale_exp: int) -> torch.Tensor:
    lib = _lib.load()
    _need_cuda(x, torch.float32, "x")
    out = torch.empty_like(x)
    _lib.check(lib.dgp_h2_to_f32(_ptr(x), x.numel(), int(scale_exp), _ptr(out), _stream(x.device)), "dgp_h2_to_f32")
    return out


def conv2d_h2(x_h2: torch.Tensor, x_exp: int, w_hwio: np.ndarray, stride: int = 1, rate: int = 1, pad_t: int = 0, pad_l: int = 0,
              out_hw: Optional[Tuple[int, int]] = None, scale=None, bias=None, residual: Optional[torch.Tensor] = None,
              res_stride: int = 0, res_is_h2: bool = False, res_exp: int = 0, relu: bool = False, y_is_h2: bool = True,
              y_exp: int = 0):
    """One conv layer on H2 tensors through the engine's cell kernels.  -> (y, y_absmax_slots)."""
    lib = _lib.load()
    _need_cuda(x_h2, torch.float32, "x_h2")
    N, H, W, Cin = x_h2.shape
    kh, kw, cin2, cout = w_hwio.shape
    assert cin2 == Cin
    dev = x_h2.device
    if out_hw is None:
        keh, kew = (kh - 1) * rate + 1, (kw - 1) * rate + 1
        out_hw = ((H + 2 * pad_t - keh) // stride + 1, (W + 2 * pad_l - kew) // stride + 1)
    Ho, Wo = out_hw
    wp = torch.from_numpy(pack_conv_weights(w_hwio)).to(dev)
    sc = None if scale is None else torch.as_tensor(scale, dtype=torch.float32).contiguous().to(dev)
    bi = None if bias is None else torch.as_tensor(bias, dtype=torch.float32).contiguous().to(dev)
    y = torch.empty((N, Ho, Wo, cout), dtype=torch.float32, device=dev)
    rh, rw = (residual.shape[1], residual.shape[2]) if residual is not None else (0, 0)
    d = _lib.DgpConvDesc(N, H, W, Cin, cout, kh, kw, stride, rate, pad_t, pad_l, Ho, Wo, int(relu),
                         res_stride if residual is not None else 0, rh, rw)
    rng = torch.zeros((2, ABSMAX_SLOTS), dtype=torch.float32, device=dev)
    _lib.check(lib.dgp_tensor_absmax(_ptr(wp), wp.numel(), _ptr(rng[0]), _stream(dev)), "dgp_tensor_absmax")
    cells = torch.empty(wp.numel(), dtype=torch.float32, device=dev)
    _lib.check(lib.dgp_conv2d_h2(C.byref(d), _ptr(x_h2), int(x_exp), _ptr(wp), _ptr(rng[0]), _ptr(sc), _ptr(bi), _ptr(residual),
                                 int(res_is_h2), int(res_exp), _ptr(y), int(y_is_h2), int(y_exp), _ptr(rng[1]), _ptr(cells),
                                 _stream(dev)), "dgp_conv2d_h2")
    return y, rng[1]


def f32_to_h1(x: torch.Tensor, scale_exp: int) -> torch.Tensor:
    """fp32 [..., C] (C % 8 == 0) -> H1 cells (the 16-bit tier's activation format: fp16(x * 2^exp), 16 bytes per 8 channels), returned
    as a float16 tensor of x's shape -- the cells ARE plain NHWC fp16 with a per-tensor scale."""
    _need_cuda(x, torch.float32, "x")
    out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    _lib.check(_lib.load().dgp_f32_to_h1(_ptr(x), x.numel(), int(scale_exp), _ptr(out), _stream(x.device)), "dgp_f32_to_h1")
    return out


def h1_to_f32(x: torch.Tensor, scale_exp: int) -> torch.Tensor:
    _need_cuda(x, torch.float16, "x")
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().dgp_h1_to_f32(_ptr(x), x.numel(), int(scale_exp), _ptr(out), _stream(x.device)), "dgp_h1_to_f32")
    return out


def conv2d_h1(x_h1: torch.Tensor, x_exp: int, w_hwio: np.ndarray, stride: int = 1, rate: int = 1, pad_t: int = 0, pad_l: int = 0,
              out_hw: Optional[Tuple[int, int]] = None, scale=None, bias=None, residual: Optional[torch.Tensor] = None,
              res_stride: int = 0, res_exp: int = 0, relu: bool = False, y_is_h1: bool = True, y_exp: int = 0):
    """One conv layer on H1 tensors (float16 NHWC with power-of-two scales) through the 16-bit tier's cell kernels.
    -> (y: float16 H1 cells or fp32, y_absmax_slots)."""
    lib = _lib.load()
    _need_cuda(x_h1, torch.float16, "x_h1")
    N, H, W, Cin = x_h1.shape
    kh, kw, cin2, cout = w_hwio.shape
    assert cin2 == Cin
    dev = x_h1.device
    if out_hw is None:
        keh, kew = (kh - 1) * rate + 1, (kw - 1) * rate + 1
        out_hw = ((H + 2 * pad_t - keh) // stride + 1, (W + 2 * pad_l - kew) // stride + 1)
    Ho, Wo = out_hw
    wp = torch.from_numpy(pack_conv_weights(w_hwio)).to(dev)
    sc = None if scale is None else torch.as_tensor(scale, dtype=torch.float32).contiguous().to(dev)
    bi = None if bias is None else torch.as_tensor(bias, dtype=torch.float32).contiguous().to(dev)
    y = torch.empty((N, Ho, Wo, cout), dtype=torch.float16 if y_is_h1 else torch.float32, device=dev)
    if residual is not None:
        _need_cuda(residual, torch.float16, "residual")
    rh, rw = (residual.shape[1], residual.shape[2]) if residual is not None else (0, 0)
    d = _lib.DgpConvDesc(N, H, W, Cin, cout, kh, kw, stride, rate, pad_t, pad_l, Ho, Wo, int(relu),
                         res_stride if residual is not None else 0, rh, rw)
    rng = torch.zeros((2, ABSMAX_SLOTS), dtype=torch.float32, device=dev)
    _lib.check(lib.dgp_tensor_absmax(_ptr(wp), wp.numel(), _ptr(rng[0]), _stream(dev)), "dgp_tensor_absmax")
    cells = torch.empty(wp.numel(), dtype=torch.float16, device=dev)
    _lib.check(lib.dgp_conv2d_h1(C.byref(d), _ptr(x_h1), int(x_exp), _ptr(wp), _ptr(rng[0]), _ptr(sc), _ptr(bi), _ptr(residual),
                                 int(res_exp), _ptr(y), int(y_is_h1), int(y_exp), _ptr(rng[1]), _ptr(cells), _stream(dev)),
               "dgp_conv2d_h1")
    return y, rng[1]


def chain_h2(r2_h2: torch.Tensor, r2_exp: int, src2_h2: torch.Tensor, src2_exp: int, w3cat: np.ndarray, scale3, bias3,
             w1: np.ndarray, scale1, bias1, res_mode: int, xout_exp: int, r1_exp: int, h1: bool = False):
    """The engine's chain kernel at layer level (include/dgp_hip.h, dgp_chain_h2): conv3 (+ shortcut, ReLU) of a bottleneck unit and
    conv1 of the next unit in one launch.  r2 [N, Ho, Wo, C]; src2: X [N, Ho, Wo, 4C] (res_mode 1), X [N, H, W, 4C] read at (2 ho, 2 wo)
    (res_mode 2) or the shortcut conv's input [N, Ho, Wo, CIN2] (res_mode 0; w3cat then has C + CIN2 rows).
    -> (x_out H2, r1 H2, x_out range slots, r1 range slots).  h1: every tensor is an H1 tensor (torch.float16 [N, H, W, C]; dgp_chain_h1)."""
    lib = _lib.load()
    dt = torch.float16 if h1 else torch.float32
    _need_cuda(r2_h2, dt, "r2_h2")
    _need_cuda(src2_h2, dt, "src2_h2")
    N, Ho, Wo, Cc = r2_h2.shape
    K, C4 = w3cat.shape
    C1 = w1.shape[1]
    assert C4 == 4 * Cc and w1.shape[0] == C4
    cin2 = K - Cc
    dev = r2_h2.device
    xo = torch.empty((N, Ho, Wo, C4), dtype=dt, device=dev)
    r1 = torch.empty((N, Ho, Wo, C1), dtype=dt, device=dev)
    rng = torch.zeros((2, ABSMAX_SLOTS), dtype=torch.float32, device=dev)
    f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
    w3c, s3, b3, w1c, s1, b1 = f(w3cat), f(scale3), f(bias3), f(w1), f(scale1), f(bias1)
    hp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    _lib.check((lib.dgp_chain_h1 if h1 else lib.dgp_chain_h2)(N, Ho, Wo, Cc, C1, cin2, int(res_mode), src2_h2.shape[1], src2_h2.shape[2], _ptr(r2_h2), int(r2_exp),
                                _ptr(src2_h2), int(src2_exp), hp(w3c), hp(s3), hp(b3), hp(w1c), hp(s1), hp(b1), _ptr(xo), int(xout_exp),
                                _ptr(r1), int(r1_exp), _ptr(rng[0]), _ptr(rng[1]), _stream(dev)), "dgp_chain_h2")
    return xo, r1, rng[0], rng[1]


def unit_h2(r1_h2: torch.Tensor, r1_exp: int, src2_h2: torch.Tensor, src2_exp: int, w2: np.ndarray, scale2, bias2, r2_exp: int,
            w3cat: np.ndarray, scale3, bias3, w1: np.ndarray, scale1, bias1, res_mode: int, xout_exp: int, r1out_exp: int, h1: bool = False):
    """The engine's unit kernel at layer level (include/dgp_hip.h, dgp_unit_h2): conv2 (3x3, stride 1) + conv3 (+ shortcut, ReLU) of a
    bottleneck unit and conv1 of the next unit in one launch.  r1 [N, H, W, C] is conv2's input; w2 HWIO [3, 3, C, C].
    -> (x_out H2, r1_out H2, r2 range slots, x_out range slots, r1_out range slots).  h1: H1 tensors (torch.float16; dgp_unit_h1)."""
    lib = _lib.load()
    dt = torch.float16 if h1 else torch.float32
    _need_cuda(r1_h2, dt, "r1_h2")
    _need_cuda(src2_h2, dt, "src2_h2")
    N, H, W, Cc = r1_h2.shape
    K, C4 = w3cat.shape
    C1 = w1.shape[1]
    assert C4 == 4 * Cc and w1.shape[0] == C4 and w2.shape == (3, 3, Cc, Cc)
    cin2 = K - Cc
    dev = r1_h2.device
    xo = torch.empty((N, H, W, C4), dtype=dt, device=dev)
    r1o = torch.empty((N, H, W, C1), dtype=dt, device=dev)
    rng = torch.zeros((3, ABSMAX_SLOTS), dtype=torch.float32, device=dev)
    f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
    arrs = [f(a) for a in (w2, scale2, bias2, w3cat, scale3, bias3, w1, scale1, bias1)]
    hp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    _lib.check((lib.dgp_unit_h1 if h1 else lib.dgp_unit_h2)(N, H, W, Cc, C1, cin2, int(res_mode), _ptr(r1_h2), int(r1_exp), _ptr(src2_h2), int(src2_exp),
                               hp(arrs[0]), hp(arrs[1]), hp(arrs[2]), int(r2_exp), hp(arrs[3]), hp(arrs[4]), hp(arrs[5]), hp(arrs[6]),
                               hp(arrs[7]), hp(arrs[8]), _ptr(xo), int(xout_exp), _ptr(r1o), int(r1out_exp), _ptr(rng[0]), _ptr(rng[1]),
                               _ptr(rng[2]), _stream(dev)), "dgp_unit_h2")
    return xo, r1o, rng[0], rng[1], rng[2]


def maxpool_3x3s2_same(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _need_cuda(x, torch.float32, "x")
    N, H, W, Cn = x.shape
    y = torch.empty((N, (H + 1) // 2, (W + 1) // 2, Cn), dtype=torch.float32, device=x.device)
    _lib.check(lib.dgp_maxpool_3x3s2_same(_ptr(x), N, H, W, Cn, _ptr(y), _stream(x.device)), "dgp_maxpool")
    return y


def preprocess_u8(frames: torch.Tensor, mean=MEAN_PIXEL) -> torch.Tensor:
    lib = _lib.load()
    _need_cuda(frames, torch.uint8, "frames")
    out = torch.empty(tuple(frames.shape[:-1]) + (4,), dtype=torch.float32, device=frames.device)
    m = (C.c_float * 3)(*mean)
    _lib.check(lib.dgp_preprocess_u8(_ptr(frames), frames.numel() // 3, m, _ptr(out), _stream(frames.device)),
               "dgp_preprocess_u8")
    return out


def motion_energy(frames: torch.Tensor, prev: Optional[torch.Tensor] = None) -> np.ndarray:
    """Motion energy of a device-resident uint8 frame sequence [T, ...] (DGP/dataset.py:29-43):
    me[t] = mean over the frame of (f_t - f_{t-1}) mod 256 (the reference's uint8 subtraction wraps), me[0] = 0 unless `prev`
    (the last frame of the previous chunk) is given.  Integer sums on the device (dgp_motion_energy), float64 division here:
    bit-identical to the reference's np.mean."""
    lib = _lib.load()
    _need_cuda(frames, torch.uint8, "frames")
    frames = frames.contiguous()
    T = frames.shape[0]
    fb = frames.numel() // max(T, 1)
    if T == 0:
        return np.zeros(0, dtype=np.float64)
    if prev is not None:
        _need_cuda(prev, torch.uint8, "prev")
        prev = prev.contiguous()
        if prev.numel() != fb:
            raise _lib.DgpError("motion_energy: prev has %d bytes, a frame has %d" % (prev.numel(), fb))
    sums = torch.empty(T, dtype=torch.int64, device=frames.device)
    _lib.check(lib.dgp_motion_energy(_ptr(frames), fb, T, _ptr(prev), _ptr(sums), _stream(frames.device)), "dgp_motion_energy")
    return sums.cpu().numpy().astype(np.float64) / float(fb)
