"""Drop-in counterparts of deepgraphpose/models/eval.py on the MI355X engine.

Same entry points and signatures as the reference (DGP/models/eval.py):
  setup_dgp_eval_graph(dlc_cfg, dgp_model_file, loc_ref=False, gauss_len=1, gamma=1)     :147
  estimate_pose(proj_cfg_file, dgp_model_file, video_file, output_dir, shuffle=1, ...)    :217
  export_pose_like_dlc / load_pose_from_dlc_to_dict                                      :621 / :648
  plot_dgp(video_file, output_dir='', ...)                                               :816
The TF session call `sess.run([mu_n, scmap], feed_dict={inputs: frame[None]})` (:328) is served by
`EvalSession.run`, which batches frames through the fused HIP path (dgp_infer); the per-joint likelihood
loop (:331-343) runs inside the soft-argmax kernel.  Video decode and movie rendering are untouched
third-party territory (moviepy) and only used when installed.
"""
from __future__ import annotations

import os
import time
from os.path import join
from pathlib import Path
from typing import Dict, Optional

import numpy as np
import yaml


class _Fetch:
    """Symbolic handle standing in for a TF tensor of the eval graph."""

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "<dgp fetch %s>" % self.name


class EvalSession:
    """What `TF.Session` + the restored graph were in the reference: holds the weights and one DGPNet per
    frame size (the TF placeholder was [1, None, None, 3]); `run` accepts the same fetch list."""

    def __init__(self, weights: Dict[str, np.ndarray], depth: int, nj: int, loc_ref: bool, gauss_len, gamma,
                 mean_pixel, max_batch: int = 32, device: int = 0, tier: Optional[str] = None):
        self.weights, self.depth, self.nj, self.loc_ref = weights, depth, nj, loc_ref
        self.tier = resolve_tier(tier)
        self.gauss_len, self.gamma, self.mean_pixel = gauss_len, gamma, tuple(mean_pixel)
        self.max_batch, self.device = max_batch, device
        self._nets = {}
        self.mu_n, self.softmax_tensor = _Fetch("mu_n"), _Fetch("softmax_tensor")
        self.scmap, self.inputs = _Fetch("scmap"), _Fetch("inputs")
        self.locref = _Fetch("locref") if loc_ref else None
        self.likelihood, self.idx = _Fetch("likelihood"), _Fetch("mu_likelihoods")

    def net_for(self, h: int, w: int):
        """The engine at frame size h x w.  ONE net is kept: a new size re-plans the layers (dgp_net_set_input_size) and
        keeps the uploaded, repacked weights -- the TF placeholder was [1, None, None, 3] and accepted any size too."""
        from .. import engine
        net = self._nets.get("net")
        if net is not None and (net.max_batch < self.max_batch or net.device.index != int(self.device)):
            self._nets = {}                        # (a kept session asked for larger batches / another GPU: new engines)
            net = None
        if net is None:
            net = engine.DGPNet(self.depth, self.nj, h, w, max_batch=self.max_batch, with_locref=self.loc_ref,
                                device=self.device, mean_pixel=self.mean_pixel, tier=self.tier)
            net.load_weights(self.weights)
            self._nets["net"] = net
        elif (net.in_h, net.in_w) != (h, w):
            net.set_input_size(h, w)
        return net

    def pipe_for(self, h: int, w: int, n_streams: int = 2):
        """The streaming form of net_for: engine.DGPPipeline (n engines on n HIP streams, batches dealt in turn) whose engine 0 is
        net_for's engine; the further engines load the same weights once."""
        from .. import engine
        net = self.net_for(h, w)
        pipe = self._nets.get("pipe")
        if pipe is None:
            pipe = engine.DGPPipeline(self.depth, self.nj, h, w, max_batch=self.max_batch, with_locref=self.loc_ref,
                                      device=self.device, n_streams=n_streams, mean_pixel=self.mean_pixel, first=net, tier=self.tier)
            for n in pipe.nets[1:]:
                n.load_weights(self.weights)
            self._nets["pipe"] = pipe
        elif (pipe.in_h, pipe.in_w) != (h, w):
            pipe.set_input_size(h, w)
        return pipe

    def run(self, fetches, feed_dict):
        import torch
        from .. import engine
        single = not isinstance(fetches, (list, tuple))
        fl = [fetches] if single else list(fetches)
        frames = feed_dict[self.inputs]
        if isinstance(frames, np.ndarray):
            if frames.dtype != np.uint8:
                # the reference feeds img_as_ubyte frames cast to fp32 (eval.py:326-328): integral 0..255
                frames = np.clip(np.rint(frames), 0, 255).astype(np.uint8)
            frames = torch.from_numpy(np.require(frames, requirements=["C", "W"])).cuda(self.device)      # (read-only memmaps: copy)
        B, h, w, _ = frames.shape
        net = self.net_for(h, w)
        names = {f.name for f in fl}
        out = {}
        for s in range(0, B, self.max_batch):
            fb = frames[s:s + self.max_batch].contiguous()
            if self.loc_ref and "locref" in names:
                scm, loc = net.forward(fb, want_locref=True)
            else:
                scm, loc = net.forward(fb), None
            part = {"scmap": scm, "locref": loc}
            if names & {"mu_n", "softmax_tensor", "likelihood", "mu_likelihoods"}:
                mu, conf, idx, pmap = engine.soft_argmax(scm, self.gamma, self.gauss_len, want_pmap=True)
                part.update(mu_n=mu, softmax_tensor=pmap, likelihood=conf, mu_likelihoods=idx)
            for k, v in part.items():
                if k in names:
                    out.setdefault(k, []).append(v.cpu().numpy())
        res = [np.concatenate(out[f.name], 0) for f in fl]
        return res[0] if single else res

    def close(self):
        """tf.Session.close(): drops the engines -- unless this is the session kept for the next call on the same snapshot
        (setup_dgp_eval_graph; clear_session_cache() frees it)"""
        if _SESSION_CACHE.get("sess") is not self:
            self._nets = {}


def resolve_tier(tier: Optional[str] = None) -> Optional[str]:
    """The arithmetic tier of an entry point: the `tier` argument, else the environment's DGP_EVAL_TIER, else None = the library's
    default (the parity tier: 1e-3 px / bit-exact indices).  "f16" = the 16-bit tier (2-byte activation cells, one MFMA per product,
    ~2 x the frames/s): a REPORTED tier with measured error, never what parity claims are made on."""
    t = tier if tier is not None else (os.environ.get("DGP_EVAL_TIER") or None)
    if t is None:
        return None
    t = str(t).lower()
    if t in ("bf16", "fp16", "half", "16", "h1"):
        t = "f16"
    if t not in ("parity", "f32x", "f16"):
        raise ValueError("tier must be 'parity' or 'f16' (got %r)" % (tier if tier is not None else os.environ.get("DGP_EVAL_TIER"),))
    return t


def setup_dgp_eval_graph(dlc_cfg, dgp_model_file, loc_ref=False, gauss_len=1, gamma=1, tier=None):
    """-> (sess, mu_n, softmax_tensor, scmap, locref, inputs), as eval.py:147-214.  `tier` is new (resolve_tier): None / "parity" / "f16".

    `dgp_model_file` is what Saver.restore takes (eval.py:194-211): the prefix of a TF V2 bundle (`<prefix>.index` +
    `.data-*`, written by the reference or by this package's fit drivers), a V1 `.ckpt` file, or an .npz / .safetensors
    file with TF variable names; optimiser slots and non-float variables (global_step) in a checkpoint are skipped;
    a missing file raises FileNotFoundError, a net_type that does not match the snapshot raises
    KeyError (the reference relies on exactly that failure to fall back from resnet_50 to resnet_101)."""
    from .. import weights_io
    depth = int(str(dlc_cfg.net_type).split("_")[-1])
    mean_pixel = dlc_cfg.get("mean_pixel", [123.68, 116.779, 103.939])
    # One session is kept between calls (run_dgp_demo / plot_dgp label a project's videos one after the other with ONE snapshot: the
    # reference restored the graph for every video): same snapshot files (path, size, mtime), same graph arguments -> the engines with
    # their uploaded, re-packed weights are reused; activation scales are calibrated again on every video's first batch.
    key = None
    if os.environ.get("DGP_EVAL_SESSION_CACHE", "1") != "0":
        try:
            f = weights_io.resolve(str(dgp_model_file))
            stamp = tuple((q, os.path.getsize(q), os.stat(q).st_mtime_ns) for q in sorted(_snapshot_files(f)))
            key = (stamp, depth, int(dlc_cfg.num_joints), bool(loc_ref), gauss_len, gamma, tuple(float(v) for v in mean_pixel), resolve_tier(tier))
        except OSError:
            key = None
    if key is not None and _SESSION_CACHE.get("key") == key:
        sess = _SESSION_CACHE["sess"]
        return sess, sess.mu_n, sess.softmax_tensor, sess.scmap, sess.locref, sess.inputs
    weights = weights_io.load_weights(str(dgp_model_file))
    if ("resnet_v1_%d/conv1/weights" % depth) not in weights:
        raise KeyError("snapshot %s holds no resnet_v1_%d variables" % (dgp_model_file, depth))
    sess = EvalSession(weights, depth, int(dlc_cfg.num_joints), bool(loc_ref), gauss_len, gamma, mean_pixel, tier=tier)
    if key is not None:
        clear_session_cache()
        _SESSION_CACHE.update(key=key, sess=sess)
    return sess, sess.mu_n, sess.softmax_tensor, sess.scmap, sess.locref, sess.inputs


_SESSION_CACHE: Dict[str, object] = {}


def _snapshot_files(resolved: str):
    """the files a snapshot consists of: a V2 bundle's .index + .data-* shards, or the single file"""
    import glob
    if resolved.endswith(".index"):
        return [resolved] + glob.glob(resolved[:-len(".index")] + ".data-*")
    return [resolved]


def clear_session_cache():
    """Drop the engines kept for the next call on the same snapshot (frees their device memory)."""
    old = _SESSION_CACHE.pop("sess", None)
    _SESSION_CACHE.pop("key", None)
    if old is not None:
        old._nets = {}


# counters of the last estimate_pose call (tests, soak runs): chunks processed and chunks re-run after a range overflow
RUN_STATS = {"chunks": 0, "chunk_reruns": 0, "strict_passes": 0, "stage_s": 0.0, "wait_frames_s": 0.0, "wait_h2d_s": 0.0, "drain_s": 0.0,
             "setup_s": 0.0, "alloc_s": 0.0, "calibrate_s": 0.0, "finish_s": 0.0}

# The host pipeline's geometry (measured settings, not run-time switches; tests patch the module attributes): engines / HIP streams the
# batches are dealt to, pinned staging slots, host threads staging an in-memory stack, copy streams the uploads alternate on, bytes of
# frames a chunk keeps resident for a re-run after a range overflow.
EVAL_STREAMS, PINNED_SLOTS, STAGE_THREADS, COPY_STREAMS, CHUNK_BYTES = 2, 8, 2, 2, 1 << 30

# pinned staging ring, kept between calls (pinning host memory costs ~ 10 ms per 30-MB buffer; a project's videos share one frame size)
_PINNED = {"key": None, "bufs": []}


def _pinned_ring(nslots: int, shape):
    import torch
    key = (nslots, tuple(shape))
    if _PINNED["key"] != key:
        _PINNED["bufs"] = []                       # (drop the old ring first)
        _PINNED["bufs"] = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(nslots)]
        _PINNED["key"] = key
    return _PINNED["bufs"]


def estimate_pose(proj_cfg_file, dgp_model_file, video_file, output_dir, shuffle=1, save_pose=True, save_str="",
                  new_size=None, crop_size=None, batch_size: int = 32, tier: Optional[str] = None):
    """Estimate pose on an arbitrary video (eval.py:217-372).  Returns {'x','y','likelihoods'} [T,nj] float64,
    or the csv path if labels already exist (:247-249).  `batch_size` is new: frames go through the GPU in
    batches instead of one sess.run per frame.  `tier` is new (resolve_tier): None = DGP_EVAL_TIER or the parity tier; "f16" = the
    16-bit tier (reported error band, ~2 x the frames/s).

    Multi-GPU: under torchrun (one process per GPU; RANK / WORLD_SIZE / LOCAL_RANK in the environment, or an
    already initialised torch.distributed group) rank r decodes and infers only the contiguous frame block
    shard_range(T, r, W), ONE RCCL all-gather of the packed keypoints reassembles the [T, nj] trajectory on every rank, and rank 0
    writes the csv / h5.  Every rank returns the full label dict."""
    from ..config import get_train_config
    from ..frames import open_frame_source
    from .. import dist as ddist
    import torch.distributed as tdist
    t_entry = time.perf_counter()

    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not tdist.is_initialized():
        ddist.init_from_env()                      # before anything touches the GPU
    world = tdist.get_world_size() if tdist.is_initialized() else 1
    rank = tdist.get_rank() if tdist.is_initialized() else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0
    if world > 1:          # the decode / staging threads started below inherit the mask: each rank's host work stays on its GPU's NUMA node
        ddist.bind_to_gpu_numa_node(local_rank)

    f = os.path.basename(str(video_file)).rsplit(".", 1)
    save_file = join(output_dir, f[0] + "_labeled%s" % save_str)
    if ddist.from_rank0(os.path.exists(save_file + ".csv")):      # rank 0 decides for everyone: a per-rank test could diverge and leave
        print("labels already exist! video at %s will not be processed" % video_file)      # some ranks alone in the collectives below
        return save_file + ".csv"

    video_clip = open_frame_source(video_file)
    n_frames = int(video_clip.n_frames)
    with open(proj_cfg_file, "r") as stream:
        proj_config = yaml.safe_load(stream)
    proj_config["video_path"] = None
    dlc_cfg = get_train_config(proj_config, shuffle=shuffle)

    try:
        dlc_cfg.net_type = "resnet_50"
        sess, mu_n, _, scmap, _, inputs = setup_dgp_eval_graph(dlc_cfg, dgp_model_file, tier=tier)
    except KeyError:
        dlc_cfg.net_type = "resnet_101"
        sess, mu_n, _, scmap, _, inputs = setup_dgp_eval_graph(dlc_cfg, dgp_model_file, tier=tier)
    sess.max_batch = int(batch_size)
    sess.device = local_rank
    lo, hi = ddist.shard_range(n_frames, rank, world)        # this rank's frames
    n_local = hi - lo

    nj = dlc_cfg.num_joints
    markers = np.zeros((n_frames, nj, 2))
    likelihoods = np.zeros((n_frames, nj))
    scale_x = scale_y = 1

    def prep(frame):
        nonlocal scale_x, scale_y
        if new_size is None and crop_size is None:
            return np.asarray(frame)
        from PIL import Image
        im = Image.fromarray(frame)
        if new_size is not None:
            scale_x = im.width / new_size[1]
            scale_y = im.height / new_size[0]
            im = im.resize(size=(new_size[1], new_size[0]))
        if crop_size is not None:
            im = im.crop(crop_size)
        return np.asarray(im)

    def _infer_once(video_clip, first_pass=True):
        nonlocal net_used
        # Host pipeline: a decode thread fills pinned staging buffers, a copy stream moves batch k+1
        # to the GPU while batch k runs through dgp_infer on the compute stream, and the keypoints of the whole video
        # come back in ONE device-to-host copy at the end (the reference fetched the full scoremap every frame).
        import itertools
        import queue
        import threading
        import torch
        dev = torch.device("cuda", sess.device)
        if world > 1 and hasattr(video_clip, "frame_at"):          # a shard starts in the middle of the video: seek
            frames_it = (video_clip.frame_at(t) for t in range(lo, hi))
            first = video_clip.frame_at(lo if n_local > 0 else 0)
            if n_local > 0:
                next(frames_it)
        else:
            frames_it = iter(video_clip.iter_frames())
            first = next(frames_it, None)
            for _ in range(lo):                                    # sources without random access: decode up to the shard
                first = next(frames_it, None)
        if first is None:
            raise ValueError("no frames in %s" % video_file)
        f0 = prep(first)
        hh, ww = f0.shape[:2]
        # two engines on two HIP streams, batches dealt in turn (engine.DGPPipeline)
        t_ = time.perf_counter()
        net = net_used = sess.pipe_for(hh, ww, n_streams=EVAL_STREAMS)
        torch.cuda.synchronize(dev)
        RUN_STATS["setup_s"] += time.perf_counter() - t_      # the engines of this frame size: weights re-packed and uploaded (first call of a size)
        t_ = time.perf_counter()
        nslots = PINNED_SLOTS                     # pinned staging ring: batches being staged / copied + slack for bursts
        pinned = _pinned_ring(nslots, (batch_size, hh, ww, 3))
        # The frames of a CHUNK of batches stay on the device until the chunk's range check has come back clean: a chunk whose
        # activations outgrew the calibrated H2 scales is re-run from HBM, without decoding anything again (DGP_EVAL_CHUNK_BATCHES,
        # default 64 batches, capped at 4 GB of frames)
        # -- never more batches than the shard holds, and at most CHUNK_BYTES (1 GiB) of frames: 34 batches of 32 at
        # 640 x 480, 24 batches of 16 at 1280 x 720, next to the engines' workspaces.  Every term is the same on every rank (the chunk
        # rounds below are collective), so nothing here may depend on a rank's free memory.
        batch_bytes = batch_size * hh * ww * 3
        per_rank_batches = max(1, -(-(-(-n_frames // world)) // batch_size))
        chunk_cap = int(CHUNK_BYTES // max(batch_bytes, 1)) or 1
        chunk_batches = max(1, min(int(os.environ.get("DGP_EVAL_CHUNK_BATCHES", "64")), chunk_cap, per_rank_batches))
        dchunk = torch.empty((chunk_batches, batch_size, hh, ww, 3), dtype=torch.uint8, device=dev)
        RUN_STATS["alloc_s"] += time.perf_counter() - t_        # pinned ring (kept between calls) + the chunk's device buffer
        strict = os.environ.get("DGP_EVAL_STRICT", "0") == "1"
        stale = False                             # an earlier chunk holds results of narrower scales than the video ended with
        cal_batch = None                          # the batch every engine (and every rank) calibrates its activation scales on
        if first_pass:
            net.reset_scales()                    # engines kept from an earlier video: THIS video's first batch sets the scales, on the default
                                                  # headroom -- the same bits as a fresh session (a strict re-pass keeps the widened scales)
        if world > 1 and hasattr(video_clip, "frame_at"):
            # every rank calibrates the activation scales on the video's FIRST batch (not on its own shard's), so the frozen scales --
            # and with them every output bit -- are those of a single-process run
            nb0 = min(batch_size, n_frames)
            for t in range(nb0):
                np.copyto(pinned[0][t].numpy(), prep(video_clip.frame_at(t)))
            cal_batch = pinned[0][:nb0].to(dev)
            net.calibrate(cal_batch, sess.gamma, sess.gauss_len)
            torch.cuda.synchronize(dev)
        # ---- staging: N host threads fill the pinned ring (a batch is ONE GIL-free copy when the source is an in-memory stack; decoders
        # are sequential by nature and keep one thread), the consumer below takes the batches IN ORDER.  Protocol, all under `cv`:
        # staged[k] = (slot, frames) of batch k; a thread may stage batch k only while k < n_freed + nslots (the window of batches that can
        # hold a slot at once: no thread can starve an earlier batch of its slot); total = number of batches once known; err = a decode error.
        cv = threading.Condition()
        st = {"staged": {}, "n_freed": 0, "total": None, "err": None, "free": list(range(nslots))}
        whole_batches = new_size is None and crop_size is None and hasattr(video_clip, "iter_batches") and hasattr(video_clip, "frames")
        n_stage = STAGE_THREADS if whole_batches else 1

        def take_slot(k):
            with cv:
                while not (k < st["n_freed"] + nslots and st["free"]) and st["err"] is None:
                    cv.wait()
                if st["err"] is not None:
                    raise RuntimeError("staging stopped")
                return st["free"].pop()

        def publish(k, slot, nb):
            with cv:
                st["staged"][k] = (slot, nb)
                cv.notify_all()

        def stage_stack(tid):                     # in-memory stack: thread tid stages batches tid, tid + n_stage, ...
            try:
                src = video_clip.frames
                nbat = -(-n_local // batch_size)
                for k in range(tid, nbat, n_stage):
                    a = lo + k * batch_size
                    nb = min(batch_size, hi - a)
                    slot = take_slot(k)
                    t_ = time.perf_counter()
                    np.copyto(pinned[slot][:nb].numpy(), src[a:a + nb])
                    RUN_STATS["stage_s"] += time.perf_counter() - t_
                    publish(k, slot, nb)
            except BaseException as e:            # surface errors in the consumer
                with cv:
                    st["err"] = st["err"] or e
                    cv.notify_all()

        def stage_decoded():                      # frame by frame from the decoder (one thread: decoding is sequential)
            try:
                k, fill, count, slot = 0, 0, 0, None
                for fr in itertools.chain([f0], (prep(x) for x in frames_it)):
                    if count >= n_local:
                        break
                    if slot is None:
                        slot = take_slot(k)
                    np.copyto(pinned[slot][fill].numpy(), fr)
                    fill += 1
                    count += 1
                    if fill == batch_size:
                        publish(k, slot, fill)
                        k, fill, slot = k + 1, 0, None
                if fill:
                    publish(k, slot, fill)
                    k += 1
                with cv:
                    st["total"] = k
                    cv.notify_all()
            except BaseException as e:
                with cv:
                    st["err"] = st["err"] or e
                    cv.notify_all()

        if whole_batches:
            st["total"] = -(-n_local // batch_size)
            threads = [threading.Thread(target=stage_stack, args=(t,), daemon=True) for t in range(n_stage)]
        else:
            threads = [threading.Thread(target=stage_decoded, daemon=True)]
        for th in threads:
            th.start()
        copy_streams = [torch.cuda.Stream(device=dev) for _ in range(COPY_STREAMS)]
        compute = torch.cuda.current_stream(dev)
        traj = torch.zeros((max(n_local, 1), nj, 5), dtype=torch.float32, device=dev)      # packed (row, col, likelihood, iy, ix)
        pending = []                              # (H2D-complete event, pinned slot): slots go back to the ring without the host waiting for every copy

        def release(block):
            """pinned slots whose H2D copy has completed go back to the ring; block: wait for the oldest copy when none has"""
            freed = 0
            while pending and pending[0][0].query():
                freed += 1
                slot = pending.pop(0)[1]
                with cv:
                    st["free"].append(slot); st["n_freed"] += 1
                    cv.notify_all()
            if block and not freed and pending:
                t_ = time.perf_counter()
                pending[0][0].synchronize()
                RUN_STATS["wait_h2d_s"] += time.perf_counter() - t_
                release(False)

        def next_batch(k):
            """(slot, frames) of batch k, or None when the shard has no batch k"""
            t_ = time.perf_counter()
            try:
                while True:
                    with cv:
                        if st["err"] is not None:
                            raise st["err"]
                        if k in st["staged"]:
                            return st["staged"].pop(k)
                        if st["total"] is not None and k >= st["total"]:
                            return None
                        if not pending:
                            cv.wait(0.05)
                            continue
                    release(True)                 # the producers may be waiting for a slot this thread still holds
            finally:
                RUN_STATS["wait_frames_s"] += time.perf_counter() - t_

        try:
            start, finished, kb = 0, False, 0
            # every rank runs the SAME number of chunk rounds (a short shard ends with empty ones): the decision to re-calibrate after a
            # range overflow is a collective
            per_rank = -(-n_frames // world)
            n_rounds = max(1, -(-(-(-per_rank // batch_size)) // chunk_batches))
            for rnd in range(n_rounds):
                entries = []                          # (slot in dchunk, frames, offset in traj) of this chunk
                while len(entries) < chunk_batches and not finished:
                    item = next_batch(kb)
                    if item is None:
                        finished = True
                        break
                    kb += 1
                    slot, nb = item
                    k = len(entries)
                    cs = copy_streams[kb % len(copy_streams)]
                    with torch.cuda.stream(cs):
                        dchunk[k][:nb].copy_(pinned[slot][:nb], non_blocking=True)
                        copied = torch.cuda.Event()
                        copied.record(cs)
                    compute.wait_event(copied)
                    pending.append((copied, slot))
                    if cal_batch is None:
                        cal_batch = dchunk[k][:nb].clone()
                        t_ = time.perf_counter()
                        net.calibrate(cal_batch, sess.gamma, sess.gauss_len)      # (what the first submit would do: timed apart)
                        RUN_STATS["calibrate_s"] += time.perf_counter() - t_
                    # written in place by the soft-argmax kernel, on the next engine's stream
                    net.submit(dchunk[k][:nb], traj[start:start + nb], sess.gamma, sess.gauss_len)
                    entries.append((k, nb, start))
                    start += nb
                    release(len(pending) >= nslots - 1)
                # H2 activation scales (include/dgp_hip.h): a batch that outgrew the scales calibrated on the first batch invalidates the
                # results since the last clean check, i.e. THIS chunk's.  All ranks decide together; every engine of every rank then
                # re-calibrates on the calibration batch with 3 more bits of headroom (same scales everywhere again) and the ranks whose
                # chunk overflowed re-run it from the frames still resident in HBM.
                for attempt in range(5):
                    t_ = time.perf_counter()
                    net.join()
                    torch.cuda.synchronize(dev)
                    RUN_STATS["drain_s"] += time.perf_counter() - t_
                    overflow = bool(net.range_status()[0])
                    anywhere = ddist.any_rank(overflow, device="cuda:%d" % sess.device)
                    if not anywhere:
                        break
                    if attempt == 4:
                        raise RuntimeError("activation scales did not settle after 4 re-calibrations in %s" % video_file)
                    if not overflow:
                        net.widen()                   # follow the rank that overflowed: same headroom everywhere
                    net.calibrate(cal_batch, sess.gamma, sess.gauss_len)
                    if rnd > 0:
                        stale = True                  # chunks [0, rnd) were computed with the narrower scales (valid, but other bits)
                    if overflow or strict:            # strict: every rank re-runs this chunk on the new scales, not only the one that overflowed
                        print("activation ranges outgrew the calibrated scales: re-calibrated, re-running frames %d-%d of %s"
                              % (lo + entries[0][2] if entries else lo, lo + start, video_file), flush=True)
                        RUN_STATS["chunk_reruns"] += 1
                        for k, nb, off in entries:
                            net.submit(dchunk[k][:nb], traj[off:off + nb], sess.gamma, sess.gauss_len)
                RUN_STATS["chunks"] += 1 if entries else 0
            if not finished:
                assert next_batch(kb) is None, "more frames than the shard holds"
            while pending:
                release(True)
            for th in threads:
                th.join()
        except BaseException as e:               # a failure on this side must not leave the staging threads waiting for a slot for ever
            with cv:
                st["err"] = st["err"] or e
                cv.notify_all()
            raise
        t_ = time.perf_counter()
        if world > 1:                                  # ONE all-gather per video: 20 bytes per (frame, joint)
            full = ddist.gather_trajectory(traj[:n_local], n_frames)
            mu_t, lik_t, _ = ddist.unpack_keypoints(full)
            markers[:] = mu_t.cpu().numpy()
            likelihoods[:] = lik_t.cpu().numpy()
        else:
            mu_t, lik_t, _ = ddist.unpack_keypoints(traj[:start])
            markers[:start] = mu_t.cpu().numpy()
            likelihoods[:start] = lik_t.cpu().numpy()
        RUN_STATS["finish_s"] += time.perf_counter() - t_       # the trajectory's gather / ONE device-to-host copy
        return stale

    # Bit-identity of a sharded run with a single-process run holds as long as no chunk overflows (or only the first one does).  After an
    # overflow in a LATER chunk the earlier chunks keep the (valid) results of the narrower scales, which a run that started with the
    # wide scales would not reproduce bit for bit.  DGP_EVAL_STRICT=1 restores the guarantee at the cost of a second pass over the video:
    # the engines keep the widened headroom and everything is computed again on those scales (the decision is collective).
    net_used = None
    RUN_STATS["chunks"] = RUN_STATS["chunk_reruns"] = RUN_STATS["strict_passes"] = 0
    # where the host side of the call spent its time: staging copies of in-memory frames (producer thread), the consumer waiting for a
    # decoded batch / for an H2D copy / for the engines at a chunk boundary
    RUN_STATS["stage_s"] = RUN_STATS["wait_frames_s"] = RUN_STATS["wait_h2d_s"] = RUN_STATS["drain_s"] = 0.0
    RUN_STATS["alloc_s"] = RUN_STATS["calibrate_s"] = RUN_STATS["finish_s"] = 0.0
    RUN_STATS["setup_s"] = time.perf_counter() - t_entry      # config, snapshot -> engine (read, re-pack, upload): before the first frame moves
    for _pass in range(4):
        if not (_infer_once(video_clip, _pass == 0) and os.environ.get("DGP_EVAL_STRICT", "0") == "1"):
            break
        RUN_STATS["strict_passes"] += 1
        print("DGP_EVAL_STRICT: scales were widened after the first chunk; computing %s again on the final scales" % video_file, flush=True)
    sess.close()
    video_clip.close()

    xr = markers[:, :, 1] * dlc_cfg.stride + 0.5 * dlc_cfg.stride      # eval.py:352-353
    yr = markers[:, :, 0] * dlc_cfg.stride + 0.5 * dlc_cfg.stride
    xr *= scale_x
    yr *= scale_y
    labels = {"x": xr, "y": yr, "likelihoods": likelihoods}
    if save_pose and rank == 0:
        if not Path(save_file).parent.exists():
            os.makedirs(os.path.dirname(save_file))
        export_pose_like_dlc(labels, os.path.basename(str(dgp_model_file)), dlc_cfg.all_joints_names, save_file)
    if world > 1:
        tdist.barrier()                # the csv / h5 is complete before any rank goes on (callers read it back)
    return labels


def export_pose_like_dlc(labels, scorer, joints_names, save_file):
    """DLC-format export (eval.py:621-645): columns MultiIndex (scorer, bodyparts, coords), index = frame
    number; csv always, hdf5 (key df_with_missing, table format) when pytables is installed."""
    import pandas as pd
    n_frames, n_labels = labels["x"].shape
    data = np.empty((n_frames, 3 * n_labels), dtype=labels["x"].dtype)
    data[:, 0::3] = labels["x"]
    data[:, 1::3] = labels["y"]
    data[:, 2::3] = labels["likelihoods"]
    cols = pd.MultiIndex.from_product([[scorer], list(joints_names), ["x", "y", "likelihood"]],
                                      names=["scorer", "bodyparts", "coords"])
    df = pd.DataFrame(data, columns=cols, index=np.arange(n_frames))
    try:
        df.to_hdf(save_file + ".h5", key="df_with_missing", format="table", mode="w")
    except ImportError:
        print("pytables not installed: skipping %s.h5 (csv is written)" % save_file)
    df.to_csv(save_file + ".csv")


def load_pose_from_dlc_to_dict(filename):
    """eval.py:648-653."""
    dlc = np.genfromtxt(filename, delimiter=",", dtype=None, encoding=None)
    dlc = dlc[3:, 1:].astype("float")
    return {"x": dlc[:, 0::3], "y": dlc[:, 1::3], "likelihoods": dlc[:, 2::3]}


def plot_dgp(video_file, output_dir="", label_dir=None, proj_cfg_file=None, dgp_model_file=None, shuffle=1, dotsize=3,
             colormap="jet", save_str="", mask_threshold=0.1, new_size=None, tier=None):
    """eval.py:816-874 (`tier` is new: estimate_pose's).  Exports the labels when missing, then hands (clip, x, y, mask) to the movie
    renderer.  Drawing the annotated movie is moviepy / matplotlib work outside this package's scope: when
    those are absent the labels are still produced and the csv path is returned."""
    f = os.path.basename(str(video_file)).rsplit(".", 1)
    save_file = join(output_dir, f[0] + "_labeled%s.mp4" % save_str)
    if label_dir is None:
        label_dir = output_dir
    label_file = join(label_dir, f[0] + "_labeled%s.csv" % save_str)
    from .. import dist as ddist
    import torch.distributed as tdist
    labels = None
    if not ddist.from_rank0(os.path.exists(label_file)):
        labels = estimate_pose(proj_cfg_file, dgp_model_file, video_file, label_dir, shuffle=shuffle, save_str=save_str,
                               new_size=new_size, tier=tier)
    if not isinstance(labels, dict):                       # labels were there already (estimate_pose returns the csv path then)
        labels = load_pose_from_dlc_to_dict(label_file)
    mask_array = labels["likelihoods"].T > mask_threshold
    if tdist.is_available() and tdist.is_initialized() and tdist.get_rank() != 0:
        return label_file                                  # sharded run: rank 0 alone renders the movie
    try:
        from moviepy.editor import VideoFileClip  # noqa: F401
    except ImportError:
        print("moviepy not installed: labels written to %s, annotated movie skipped (%d/%d markers above the "
              "%.2f likelihood mask)" % (label_file, int(mask_array.sum()), mask_array.size, mask_threshold))
        return label_file
    from .render import create_annotated_movie           # thin moviepy wrapper, optional
    create_annotated_movie(video_file, labels["x"].T, labels["y"].T, mask_array=mask_array, filename=save_file,
                           dotsize=dotsize, colormap=colormap)
    return save_file


def pairwisedistances(DataCombined, scorer1, scorer2, pcutoff=-1, bodyparts=None):
    """Per-(image, bodypart) Euclidean px distance between two scorers (PET/evaluate.py:22-32)."""
    mask = DataCombined[scorer2].xs("likelihood", level=1, axis=1) >= pcutoff
    a, b = (DataCombined[scorer1], DataCombined[scorer2]) if bodyparts is None else \
        (DataCombined[scorer1][bodyparts], DataCombined[scorer2][bodyparts])
    sq = (a - b) ** 2
    rmse = np.sqrt(sq.xs("x", level=1, axis=1) + sq.xs("y", level=1, axis=1))
    return rmse, rmse[mask]


def _read_collected_data(folder, scorer):
    """CollectedData_<scorer>.h5 (key df_with_missing) when pytables is installed, else the .csv twin
    (3 header rows: scorer / bodyparts / coords; index = image path)."""
    import pandas as pd
    base = join(folder, "CollectedData_" + scorer)
    try:
        return pd.read_hdf(base + ".h5", "df_with_missing")
    except (ImportError, FileNotFoundError):
        return pd.read_csv(base + ".csv", header=[0, 1, 2], index_col=0)


def soft_argmax_locref_pose(locref, softmax_map, stride, locref_stdev):
    """eval.py:757-786 for one frame: locref [H,W,2nj] (raw head output), softmax_map [H,W,nj] (normalised map of
    argmax_2d_from_cm) -> pose [nj,3] = (soft-argmax position in px + sum_map(softmax * locref * stdev))[::-1], likelihood 1.
    The reference builds `pose_hard_st` as well and keeps `pose_hard_st1`; with a normalised map the two agree."""
    H, W = locref.shape[:2]
    nj = softmax_map.shape[-1]
    lr = np.reshape(np.asarray(locref), (H, W, -1, 2)) * locref_stdev
    xg, yg = np.meshgrid(np.linspace(0, H - 1, H), np.linspace(0, W - 1, W))
    alpha = np.array([xg, yg]).swapaxes(1, 2)                     # 2 x H x W: (row, col) grids
    out = []
    for j in range(nj):
        st_j = np.expand_dims(softmax_map[:, :, j], 0)
        lr_j = np.transpose(lr[:, :, j, :], [2, 0, 1])
        soft = np.sum(np.sum(st_j * alpha, 1), 1) * stride + 0.5 * stride
        offset = np.sum(np.sum(st_j * lr_j, 1), 1)
        out.append(np.hstack((soft + offset)[::-1]))
    return np.hstack((np.array(out), np.ones((nj, 1))))


def evaluate_dgp(proj_cfg_file, dgp_model_file, shuffle=1, loc_ref=None, loc_ref_calc="dlc", tier=None):
    """Evaluate a model by RMSE (px) on the human-labeled train/test images (eval.py:656-813).  `tier` is new (resolve_tier).

    loc_ref=True + loc_ref_calc='dlc': DLC hard arg-max + location refinement (HIP `hard_argmax` kernel);
    loc_ref=False: DGP soft-argmax (HIP `soft_argmax` kernel).  loc_ref=True + any other loc_ref_calc ('dgp'): soft-argmax
    position plus the softmax-weighted mean location-refinement offset (eval.py:752-786; the normalised map and the locref
    field come from the HIP kernels, the two small weighted sums are host numpy as in the reference).  Returns the RMSE
    DataFrame over all train/test data."""
    import pickle
    import pandas as pd
    from PIL import Image
    from ..config import get_train_config, GetTrainingSetFolder
    from .predict import pose_from_argmax
    from .. import engine
    import torch

    with open(proj_cfg_file, "r") as stream:
        proj_config = yaml.safe_load(stream)
    proj_config["video_path"] = None
    dlc_cfg = get_train_config(proj_config, shuffle=shuffle)
    loc_ref = dlc_cfg.location_refinement if loc_ref is None else loc_ref
    if not loc_ref:
        dlc_cfg.location_refinement = False
    try:
        dlc_cfg.net_type = "resnet_50"
        sess, mu_n, softmax_tensor, scmap_t, locref_t, inputs = setup_dgp_eval_graph(dlc_cfg, dgp_model_file, loc_ref=loc_ref, tier=tier)
    except KeyError:
        dlc_cfg.net_type = "resnet_101"
        sess, mu_n, softmax_tensor, scmap_t, locref_t, inputs = setup_dgp_eval_graph(dlc_cfg, dgp_model_file, loc_ref=loc_ref, tier=tier)

    tsfolder = GetTrainingSetFolder(proj_config)
    scorer_dgp = "DGP"
    Data = _read_collected_data(join(proj_config["project_path"], str(tsfolder)), proj_config["scorer"])
    bodyparts = list(proj_config["bodyparts"])
    meta = join(proj_config["project_path"], str(tsfolder), "Documentation_data-" + proj_config["Task"] + "_" +
                str(int(proj_config["TrainingFraction"][0] * 100)) + "shuffle" + str(shuffle) + ".pickle")
    with open(meta, "rb") as f:
        _, trainIndices, testIndices, _ = pickle.load(f)

    nj = len(dlc_cfg["all_joints_names"])
    pred = np.ones((len(Data.index), 3 * nj))
    for i, imagename in enumerate(Data.index):
        with Image.open(join(proj_config["project_path"], imagename)) as im:
            image = np.asarray(im.convert("RGB"))
        if loc_ref and loc_ref_calc.lower() != "dlc":
            lr, st = sess.run([locref_t, softmax_tensor], feed_dict={inputs: image[None]})
            pose = soft_argmax_locref_pose(lr[0], st[0], dlc_cfg.stride, dlc_cfg.locref_stdev)
        elif loc_ref:
            net = sess.net_for(image.shape[0], image.shape[1])
            fr = torch.from_numpy(np.require(image[None], requirements=["C", "W"])).cuda(sess.device)
            scm, loc = net.forward(fr, want_locref=True)
            idx, prob, offs = engine.hard_argmax(scm, loc)
            pose = pose_from_argmax(idx[0].cpu().numpy(), prob[0].cpu().numpy(), offs[0].cpu().numpy(), dlc_cfg.stride,
                                    dlc_cfg.locref_stdev)
        else:
            mu = sess.run(mu_n, feed_dict={inputs: image[None]})
            p = mu * dlc_cfg.stride + 0.5 * dlc_cfg.stride
            pose = np.hstack([p[0, :, ::-1], np.ones((nj, 1))])
        pred[i, :] = pose.flatten()
    sess.close()

    index = pd.MultiIndex.from_product([[scorer_dgp], dlc_cfg["all_joints_names"], ["x", "y", "likelihood"]],
                                       names=["scorer", "bodyparts", "coords"])
    DataMachine = pd.DataFrame(pred, columns=index, index=Data.index.values)
    DataCombined = pd.concat([Data.T, DataMachine.T], axis=0).T
    RMSE, _ = pairwisedistances(DataCombined, proj_config["scorer"], scorer_dgp, proj_config["pcutoff"], bodyparts)
    testerror = np.nanmean(RMSE.iloc[testIndices].values.flatten())
    trainerror = np.nanmean(RMSE.iloc[trainIndices].values.flatten())
    print("Train error:", np.round(trainerror, 2), " pixels")
    print("Test error:", np.round(testerror, 2), " pixels")
    return RMSE
