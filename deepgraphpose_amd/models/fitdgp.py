"""Drop-in counterparts of deepgraphpose/models/fitdgp.py on the MI355X training engine.

  fit_dgp_labeledonly(snapshot, dlcpath, ...)   DGP/models/fitdgp.py:257-546   (step 1)
  fit_dgp(snapshot, dlcpath, ...)               DGP/models/fitdgp.py:549-845   (step 2)
  dgp_loss(data_batcher, dgp_cfg)               DGP/models/fitdgp.py:848-1144  (loss pre-computation + closure)
  fit_dlc(snapshot, dlcpath, ...)               DGP/models/fitdgp.py:53-254    (step 0: DLC baseline trainer)

One iteration = `Trainer.step` (deepgraphpose_amd/train.py) = the reference's sess.run([loss, train_op]).
Snapshots are what the reference's tf.train.Saver leaves in the train folder (fitdgp.py:239-245, 535-540, 832-839):
TensorFlow V2 bundles `snapshot-step{k}--{it}.{index,data-00000-of-00001}`, `snapshot-step{k}--0.*` and
`snapshot-step{k}-final--0.*` plus the `checkpoint` state file, written without TensorFlow (weights_io.Saver,
tf_checkpoint.py); DGP_SNAPSHOT_FORMAT=npz writes `.npz` files of the same names instead.
Host-side hooks: augmentation of the labeled frames (`aug`; deepgraphpose_amd/augment.py: imgaug's pipeline when imgaug is
installed, the numpy / scipy restatement of the same seven augmenters otherwise) and the cv2 Farneback
optical flow feeding the temporal clique (`wt > 0`; raises ImportError without OpenCV -- the loss term itself is a HIP kernel).
"""
from __future__ import annotations

import os
import time
from os import listdir
from os.path import isfile, join
from pathlib import Path
from random import randint

import numpy as np

from .. import config as dcfg
from ..dataset import MultiDataset, coord2map
from ..loss import DGPHyper
from .fitdgp_util import gen_batch

PREFETCH_DEPTH = 2          # iterations of host batches built ahead on a background thread (0: inline); a measured setting, not a switch

_VIDEO_EXT = ("avi", "mp4", "mov", "mkv")


def _video_sets(dlc_base_path: Path, cfg):
    """<project>/videos_dgp/* when present, else the project's video_sets (fitdgp.py:589-604).  Directories of
    frames and .npy stacks are accepted next to real videos (no decoder is required for them)."""
    video_path = str(dlc_base_path / "videos_dgp")
    if not os.path.exists(video_path):
        print(video_path + " does not exist!")
        return list(cfg["video_sets"])
    out = []
    for f in sorted(listdir(video_path)):
        p = join(video_path, f)
        if (isfile(p) and (any(f.find(e) > 0 for e in _VIDEO_EXT) or f.endswith(".npy"))) or os.path.isdir(p):
            out.append(p)
    return out


def _limb_statistics(labels_list, S0, stride, ws, ws_max):
    """ws_l = ws / mean non-zero limb length, ws_max_l = ws_max * max limb length, in px (fitdgp.py:875-892)."""
    nj = S0.shape[1]
    full = np.empty((0, nj, 2))
    for j in labels_list:
        if len(j) > 0:
            full = np.vstack((j, full))
    j1 = np.copy(full).swapaxes(1, 2).reshape(-1, nj)
    j1[np.isnan(j1)] = 1e10
    limb = np.matmul(j1, S0.T)
    limb[np.abs(limb) > 1e5] = 0
    limb = np.sqrt(np.sum(np.square(np.reshape(limb, [full.shape[0], 2, -1])), 1))
    limb = limb.T * stride + stride / 2
    ws_max_v = np.max(np.nan_to_num(limb), 1) * ws_max if limb.size else np.zeros(S0.shape[0])
    with np.errstate(invalid="ignore", divide="ignore"):
        mean_nz = np.true_divide(limb.sum(1), (limb != 0).sum(1)) if limb.size else np.zeros(S0.shape[0])
    return 1 / (np.nan_to_num(mean_nz) + 1e-20) * ws, ws_max_v


def dgp_loss(data_batcher, dgp_cfg):
    """-> (loss, total_loss, total_loss_visible, placeholders), the reference's return contract (fitdgp.py:848, 1130-1144):
    `loss` maps the loss-term names to evaluable handles, `placeholders` holds the same 12 keys, and
    `TrainSession(trainer, loss.graph).run([loss, train_op], feed_dict)` is the reference's `sess.run([loss, train_op], feed_dict)`
    (models/session.py).  The pre-computation of fitdgp.py:865-892 (limb statistics -> ws, ws_max) happens here; the loss terms
    themselves are HIP kernels (csrc/dgp_loss.hip) launched by the session."""
    from .session import LossGraph
    hyper = DGPHyper(ws=dgp_cfg.ws, ws_max=dgp_cfg.ws_max, wt=dgp_cfg.wt, wt_max=dgp_cfg.wt_max,
                     wn_visible=dgp_cfg.wn_visible, wn_hidden=dgp_cfg.wn_hidden, gamma=dgp_cfg.gamma,
                     gauss_len=dgp_cfg.gauss_len, lengthscale=dgp_cfg.lengthscale, lr=dgp_cfg.lr, gm2=dgp_cfg.gm2,
                     gm3=dgp_cfg.gm3, stride=dgp_cfg.stride, locref_loss_weight=dgp_cfg.locref_loss_weight,
                     locref_huber_loss=dgp_cfg.locref_huber_loss)
    if hyper.gm2 not in (0, 1, 2) or hyper.gm3 not in (0, 3):
        raise Exception("Not implemented")                        # fitdgp.py:1019, :1036
    S0 = np.asarray(data_batcher.S0, dtype=np.float64).reshape(-1, data_batcher.nj)
    ws, ws_max = _limb_statistics([d.labels for d in data_batcher.datasets], S0, dgp_cfg.stride, dgp_cfg.ws,
                                  dgp_cfg.ws_max)
    graph = LossGraph(hyper, S0, ws, ws_max, data_batcher.n_frames_total, data_batcher.n_visible_frames_total,
                      data_batcher.nj)
    return graph.loss, graph.total_loss, graph.total_loss_visible, graph.placeholders


def _feed(placeholders, images, joint_loc, lmap, lmask, addn, wt_mask, vector_field, wt, nx_out, ny_out, learning_rate, lr):
    """The reference's feed_dict (fitdgp.py:796-815), key for key."""
    vm, hm, vt = addn
    nt = images.shape[0]
    xg, yg = np.meshgrid(np.linspace(0, nx_out - 1, nx_out), np.linspace(0, ny_out - 1, ny_out))
    alpha = np.array([xg, yg]).swapaxes(1, 2)                     # 2 x nx_out x ny_out
    return {placeholders["inputs"]: images, placeholders["targets"]: joint_loc, placeholders["locref_map"]: lmap,
            placeholders["locref_mask"]: lmask, placeholders["visible_marker_pl"]: vm, placeholders["hidden_marker_pl"]: hm,
            placeholders["visible_marker_in_targets_pl"]: vt,
            placeholders["wt_batch_mask_pl"]: np.asarray(wt_mask if wt_mask is not None else np.ones(max(nt - 1, 0))),
            placeholders["vector_field_tf"]: vector_field, placeholders["nt_batch_pl"]: nt,
            placeholders["wt_batch_pl"]: np.ones(max(nt - 1, 0)) * wt, placeholders["alpha_tf"]: alpha, learning_rate: lr}


def _setup(snapshot, dlcpath, shuffle, trainingsetindex, frame_sources):
    dlc_base_path = Path(dlcpath)
    config_path = dlc_base_path / "config.yaml"
    print("config_path", config_path)
    cfg = dcfg.read_config(config_path)
    modelfolder = dcfg.GetModelFolder(cfg["TrainingFraction"][trainingsetindex], shuffle, cfg)
    train_path = dlc_base_path / modelfolder / "train"
    video_sets = _video_sets(dlc_base_path, cfg)
    print("video_sets: ", video_sets)
    S0 = dcfg.skeleton_matrix(cfg)
    data_batcher = MultiDataset(config_yaml=config_path, video_sets=video_sets, shuffle=shuffle, S0=S0,
                                sources=frame_sources)
    return data_batcher, str(train_path / snapshot)


def _dp_index(it, n):
    """Data-parallel runs (torch.distributed initialised, world W > 1): the schedule is identical on every
    rank (same seeds), iteration `it` of rank r consumes entry it*W + r, and Trainer.step averages the gradients over the
    ranks -- W windows per optimiser step.  Single process: the reference's schedule, unchanged."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return (it * dist.get_world_size() + dist.get_rank()) % n
    return it


def _make_trainer(data_batcher, init_weights, max_frames, device=0):
    import torch  # noqa: F401
    from .. import weights_io
    from ..train import Trainer
    wts = weights_io.load_weights(init_weights)
    depth = weights_io.net_depth(wts)
    nj = data_batcher.nj
    if "pose/locref_pred/block4/weights" not in wts:
        raise KeyError("snapshot %s has no pose/locref_pred head (dgp_loss trains both heads)" % init_weights)
    # DGP_TRAIN_TIER=f16: the 16-bit tier of the training step (BASELINE configs[3]'s precision; include/dgp_hip.h, dgp_trainer_set_tier);
    # default: the parity tier
    tr = Trainer(depth, nj, data_batcher.nx_in, data_batcher.ny_in, max_frames=max_frames, device=device,
                 tier=os.environ.get("DGP_TRAIN_TIER") or None)
    tr.load_weights(wts)
    return tr


class _FrameUploader:
    """Uploads a batch's frames from the prefetch thread: numpy -> one of a few pinned staging buffers -> device, asynchronously on its
    own HIP stream, so that the frames of iteration it + 1 are already resident when the GPU finishes iteration it (a 10 MB pageable
    copy at the start of every step is 0.4 ms of idle GPU).  The device tensor carries the copy's event; _frames_to_device waits on it."""

    def __init__(self, device, slots: int = 4):
        import torch
        self.device = torch.device("cuda", device) if isinstance(device, int) else device
        self.stream = torch.cuda.Stream(device=self.device)
        self.slots, self.pinned, self.turn = slots, [None] * slots, 0
        self.copied = [None] * slots              # event behind the last H2D copy issued from each staging buffer

    def __call__(self, images):
        import torch
        img = np.ascontiguousarray(images)
        if img.dtype != np.uint8:
            img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        k = self.turn
        self.turn = (k + 1) % self.slots
        if self.copied[k] is not None:
            self.copied[k].synchronize()          # the copy queued from this buffer `slots` items ago must have read it (any prefetch depth)
        if self.pinned[k] is None or self.pinned[k].numel() < img.size:
            self.pinned[k] = torch.empty(img.size, dtype=torch.uint8).pin_memory()
        stage = self.pinned[k][:img.size].view(img.shape)
        stage.numpy()[...] = img
        with torch.cuda.stream(self.stream):
            dev = stage.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        dev._dgp_ready = ev
        self.copied[k] = ev
        return dev


def _frames_to_device(trainer, images):
    """Batch images -> uint8 device frames; the net follows the batch's resolution (videos of one project may differ in size:
    the reference's placeholders are [None, None, None, 3]).  A device tensor from _FrameUploader passes through (after its copy)."""
    import torch
    if isinstance(images, torch.Tensor) and images.is_cuda:
        ev = getattr(images, "_dgp_ready", None)
        cur = torch.cuda.current_stream(images.device)
        if ev is not None:
            cur.wait_event(ev)
        images.record_stream(cur)
        if (trainer.net.in_h, trainer.net.in_w) != tuple(images.shape[1:3]):
            trainer.set_input_size(int(images.shape[1]), int(images.shape[2]))
        return images
    img = np.ascontiguousarray(images)
    if img.dtype != np.uint8:
        img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    if (trainer.net.in_h, trainer.net.in_w) != tuple(img.shape[1:3]):
        trainer.set_input_size(int(img.shape[1]), int(img.shape[2]))
    return torch.from_numpy(img).to(trainer.device)


def _locref_targets(joint_loc, nt, vis_within, nx_out, ny_out, nj, dgp_cfg):
    lt, lm = coord2map(joint_loc, nx_out, ny_out, nj, dgp_cfg.pos_dist_thresh, dgp_cfg.locref_stdev) \
        if joint_loc.shape[0] else (np.zeros((0,)), np.zeros((0,)))
    lmap = np.zeros((nt, nx_out, ny_out, nj * 2), dtype=np.float32)       # fp32 here (on the prefetch thread): the trainer's upload of
    lmask = np.zeros((nt, nx_out, ny_out, nj * 2), dtype=np.float32)      # the two maps is then a plain copy, no conversion pass
    if lm.shape[0] != 0:
        lmap[vis_within], lmask[vis_within] = lt, lm
    return lmap, lmask


def _save(saver, trainer, prefix, step, it, final, debug=""):
    """The reference's three Saver.save calls (fitdgp.py:535-540, 832-839): `<prefix>-step{k}-` at global_step it and 0, and
    `<prefix>-step{k}-final-` at 0 after the last iteration."""
    if _is_dp():
        import torch.distributed as dist
        if dist.get_rank() != 0:                 # every rank holds the same weights: one writer
            return
    w = trainer.get_weights()
    model_name = prefix + "-step" + str(step) + debug + "-"
    saver.save(w, model_name, global_step=it)
    saver.save(w, model_name, global_step=0)
    if final:
        saver.save(w, prefix + "-step" + str(step) + debug + "-final-", global_step=0)


def _augment(dgp_cfg):
    """The augmentation pipeline of the labeled frames (fitdgp.py:446-447, 735-736: build_aug(apply_prob=0.8)), or None."""
    if not dgp_cfg.aug:
        return None
    from ..augment import build_aug
    return build_aug(apply_prob=0.8)


def _is_dp():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _dp_setup():
    """Data-parallel runs (torchrun: RANK / WORLD_SIZE in the environment): join the process group BEFORE any GPU use and
    return (rank, world, local_rank).  The batch schedule must be identical on every rank, so rank 0's RNG seeds are broadcast."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return 0, 1, 0
    import random
    import torch
    import torch.distributed as dist
    from .. import dist as ddist
    rank, local, world = ddist.init_from_env()
    seed = torch.tensor([np.random.randint(0, 2 ** 31 - 1)], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        seed = seed.cuda(local)
    dist.broadcast(seed, 0)
    np.random.seed(int(seed.item()))
    random.seed(int(seed.item()))
    return rank, world, local


def _prefetched(make_item, n_items: int, depth=None):
    """make_item(0), make_item(1), ... in order, built by ONE background thread up to `depth` items ahead of the consumer: the host
    work of iteration it + 1 (frame reads, augmentation, target maps) runs while the GPU trains on iteration it.  A single producer
    keeps the order of every random draw, so a run is the same with or without prefetching (PREFETCH_DEPTH = 0: inline)."""
    import queue
    import threading
    if depth is None:
        depth = PREFETCH_DEPTH
    if depth <= 0 or n_items <= 1:
        for i in range(n_items):
            yield make_item(i)
        return
    q = queue.Queue(maxsize=depth)
    stop = threading.Event()

    def work():
        try:
            for i in range(n_items):
                item = make_item(i)
                while not stop.is_set():
                    try:
                        q.put((item, None), timeout=0.1)
                        break
                    except queue.Full:
                        pass
                if stop.is_set():
                    return
        except BaseException as e:      # surfaces in the consumer
            while not stop.is_set():
                try:
                    q.put((None, e), timeout=0.1)
                    break
                except queue.Full:
                    pass

    th = threading.Thread(target=work, name="dgp-batch-prefetch", daemon=True)
    th.start()
    try:
        for _ in range(n_items):
            item, err = q.get()
            if err is not None:
                raise err
            yield item
    finally:
        stop.set()
        th.join(timeout=5.0)


def _pretrained_checkpoint(net_type: str, dlc_cfg) -> str:
    """ImageNet backbone checkpoint `resnet_v1_<depth>.ckpt`.  The reference looks inside the installed deeplabcut
    package (fitdgp.py:101-106); here: $DGP_PRETRAINED_DIR, <package>/pretrained/, then pose_cfg.yaml's init_weights."""
    name = net_type.split("_")[0] + "_v1_" + net_type.split("_")[1] + ".ckpt"
    here = Path(__file__).resolve().parent.parent
    cands = [Path(os.environ["DGP_PRETRAINED_DIR"]) / name] if os.environ.get("DGP_PRETRAINED_DIR") else []
    cands += [here / "pretrained" / name, Path(str(dlc_cfg.get("init_weights", "")))]
    for c in cands:
        for suffix in ("", ".npz", ".index", ".safetensors"):
            if str(c) and os.path.isfile(str(c) + suffix):
                return str(c)
    raise FileNotFoundError("ImageNet checkpoint %s not found (looked in %s); set DGP_PRETRAINED_DIR or start from a "
                            "snapshot via the `snapshot` argument" % (name, ", ".join(str(c) for c in cands)))


def _fresh_heads(wts, depth, nj, location_refinement, seed=None):
    """Head variables absent from an ImageNet checkpoint, initialised like slim.conv2d_transpose does
    (pose_net.py:37-44: xavier/glorot-uniform weights over (fan_in + fan_out)/2, zero biases)."""
    rng = np.random.RandomState(seed)
    cin = wts["resnet_v1_%d/block4/unit_3/bottleneck_v1/conv3/weights" % depth].shape[-1]
    for scope, cout in (("pose/part_pred/block4", nj), ("pose/locref_pred/block4", 2 * nj)):
        if scope + "/weights" in wts:
            continue
        fan_in, fan_out = cout * 9, cin * 9                      # kernel [3,3,out,in]: fan_in = shape[-2]*9, fan_out = shape[-1]*9
        limit = np.sqrt(3.0 / ((fan_in + fan_out) / 2.0))
        wts[scope + "/weights"] = rng.uniform(-limit, limit, size=(3, 3, cout, cin)).astype(np.float32)
        wts[scope + "/biases"] = np.zeros(cout, dtype=np.float32)
    return wts


def fit_dlc(snapshot, dlcpath, shuffle=1, step=0, saveiters=1000, displayiters=100, maxiters=200000,
            trainingsetindex=0):
    """Run the DLC baseline trainer (step 0), DGP/models/fitdgp.py:53-254: one randomly scaled / cropped labeled
    image per iteration (PoseDataset), loss = sigmoid CE on binary target disks + locref Huber (pose_net.train),
    multi-step learning rate, MomentumOptimizer(0.9) without clipping; snapshots snapshot-step0-{it} and
    snapshot-step0-final--0.  `snapshot`: a 'snapshot-*' name inside the train folder, or anything else to start
    from the ImageNet resnet_v1_<depth>.ckpt."""
    import torch
    from .. import weights_io
    from ..dlc_dataset import LearningRate, PoseDataset
    from ..train import Trainer

    # data-parallel like the other fit drivers: every rank draws the SAME sample sequence (seeds broadcast by _dp_setup) and takes
    # every W-th sample, Trainer averages the gradients over the ranks, rank 0 alone writes snapshots and the learning-stats file
    rank, world, local_rank = _dp_setup()
    dlc_base_path = Path(dlcpath)
    config_path = dlc_base_path / "config.yaml"
    print("config_path", config_path)
    cfg = dcfg.read_config(config_path)
    modelfoldername = dcfg.GetModelFolder(cfg["TrainingFraction"][trainingsetindex], shuffle, cfg)
    pose_config_yaml = Path(os.path.join(cfg["project_path"], str(modelfoldername), "train", "pose_cfg.yaml"))

    dlc_cfg = dcfg.load_config(pose_config_yaml)                  # fitdgp.py:93-111: the reference's overrides
    dlc_cfg.crop = True
    dlc_cfg.cropratio = 0.4
    dlc_cfg.global_scale = 0.8
    dlc_cfg.multi_step = [[0.001, 10000], [0.005, 430000], [0.002, 730000], [0.001, 1030000]]
    final = dlc_cfg.snapshot_prefix + "-step0-final--0"
    if weights_io.exists(final):
        print(final, "  exists! The original DLC has already been run.", flush=True)
        return None
    if "snapshot" in snapshot:
        init_weights = str(dlc_base_path / modelfoldername / "train" / snapshot)
    else:
        init_weights = _pretrained_checkpoint(dlc_cfg.net_type, dlc_cfg)
    dlc_cfg.init_weights = init_weights
    dlc_cfg.pos_dist_thresh = 8
    dlc_cfg.output_stride = 16

    if dlc_cfg.batch_size != 1:
        raise ValueError("fit_dlc: batch_size must be 1 (every sample has its own size)")
    if dlc_cfg.intermediate_supervision:
        raise NotImplementedError("intermediate_supervision is off in every DGP configuration and is not built")
    if dlc_cfg.optimizer != "sgd":
        raise ValueError("unknown optimizer {}".format(dlc_cfg.optimizer))

    dataset = PoseDataset(dlc_cfg)
    wts = weights_io.load_weights(init_weights)
    depth = weights_io.net_depth(wts)
    nj = int(dlc_cfg.num_joints)
    if "snapshot" in Path(init_weights).stem:
        print("Loading already trained DLC with backbone:", dlc_cfg.net_type, flush=True)
    else:
        print("Loading ImageNet-pretrained", dlc_cfg.net_type, flush=True)
        wts = {k: v for k, v in wts.items() if k.startswith("resnet_v1")}
    wts = _fresh_heads(wts, depth, nj, dlc_cfg.location_refinement)
    trainer = Trainer(depth, nj, 64, 64, max_frames=1, device=local_rank)
    trainer.load_weights(wts)

    display_iters = max(1, int(dlc_cfg.get("display_iters", 1000) if displayiters is None else displayiters))
    save_iters = max(1, int(dlc_cfg.get("save_iters", 50000) if saveiters is None else saveiters))
    max_iter = int(dlc_cfg.multi_step[-1][1]) if maxiters is None else min(int(dlc_cfg.multi_step[-1][1]), int(maxiters))
    print("Display_iters overwritten as", display_iters, flush=True)
    print("Save_iters overwritten as", save_iters, flush=True)
    print("Max_iters overwritten as", max_iter, flush=True)

    saver = weights_io.Saver(max_to_keep=5)                       # fitdgp.py:150-152
    lr_gen = LearningRate(dlc_cfg)
    stats_path = Path(pose_config_yaml).with_name("learning_stats.csv")
    lrf = open(str(stats_path) if rank == 0 else os.devnull, "w")
    cumloss, partloss, locrefloss = 0.0, 0.0, 0.0
    print("Starting training....", flush=True)
    dev = trainer.device

    def next_sample(_i):                      # W samples per optimiser step: rank r trains on sample it * W + r of the common sequence
        batch = None
        for r in range(world):                # the other ranks' samples only advance the random streams (no image read, no target maps)
            if r == rank:
                batch = dataset.next_batch()
            else:
                dataset.skip_batch()
        return batch

    max_iter = max(1, max_iter // world) if world > 1 else max_iter
    for it, batch in enumerate(_prefetched(next_sample, max_iter + 1)):      # image read / scale / targets one step ahead
        current_lr = lr_gen.get_lr(it * world)
        img = batch["inputs"]
        trainer.set_input_size(img.shape[1], img.shape[2])
        frames = torch.from_numpy(img).to(dev)
        losses = trainer.forward_backward_dlc(
            frames, batch["part_score_targets"], batch["locref_targets"], batch["locref_mask"],
            part_score_weights=batch["part_score_weights"] if dlc_cfg.weigh_part_predictions else None,
            locref_loss_weight=dlc_cfg.locref_loss_weight, locref_huber_loss=dlc_cfg.locref_huber_loss,
            location_refinement=dlc_cfg.location_refinement)
        if world > 1:
            trainer.allreduce_gradients()         # mean gradient over the ranks (RCCL): replicas stay bit-identical
        trainer.apply_gradients(current_lr, 0.9, clip_norm=0.0)  # MomentumOptimizer, no clipping (train.py:94-113)

        partloss += losses["part_loss"]
        if dlc_cfg.location_refinement:
            locrefloss += losses["locref_loss"]
        cumloss += losses["total_loss"]
        if it % display_iters == 0 and it > 0:
            vals = (it, "total loss {0:.4f}".format(cumloss / display_iters),
                    "scoremap loss {0:.4f}".format(partloss / display_iters),
                    "learning rate {0:.4f}".format(locrefloss / display_iters), current_lr)     # labels as in fitdgp.py:212-230
            print("iteration: {} loss: {} scmap loss: {} locref loss: {} lr: {}".format(*vals), flush=True)
            lrf.write("iteration: {}, loss: {}, scmap loss: {}, locref loss: {}, lr: {}\n".format(*vals))
            lrf.flush()
        if rank == 0 and ((it % save_iters == 0 and it != 0) or it == max_iter):      # fitdgp.py:237-245
            w = trainer.get_weights()
            saver.save(w, dlc_cfg.snapshot_prefix + "-step" + str(step) + "-", global_step=it)
            if it == max_iter:
                saver.save(w, dlc_cfg.snapshot_prefix + "-step" + str(step) + "-final-", global_step=0)
    print("Finish training {} iterations\n".format(it), flush=True)
    lrf.close()
    return None


def fit_dgp_labeledonly(snapshot, dlcpath, shuffle=1, step=1, saveiters=1000, displayiters=5, maxiters=50000, ns=10,
                        nc=2048, n_max_frames=2000, aug=True, trainingsetindex=0, frame_sources=None):
    """Run DGP with labeled frames only (fitdgp.py:257-546): batch = one labeled frame, loss =
    total_loss_visible.  `frame_sources` (new, optional) injects already-open frame sources per video."""
    rank, world, local_rank = _dp_setup()
    data_batcher, init_weights = _setup(snapshot, dlcpath, shuffle, trainingsetindex, frame_sources)
    dgp_cfg = data_batcher.dlc_config
    dgp_cfg.update(ws=0, ws_max=1.2, wt=0, wt_max=0, wn_visible=1, wn_hidden=0, gamma=1, gauss_len=1, lengthscale=1,
                   max_to_keep=5, batch_size=1, n_times_all_frames=100, lr=0.005, gm2=0, gm3=0, aug=aug)
    from .. import weights_io
    final = dgp_cfg.snapshot_prefix + "-step1-final--0"
    if weights_io.exists(final):
        print(final, "  exists! DGP with labeled frames has already been run.", flush=True)
        return None
    data_batcher.create_batches_from_resnet_output(0, ns_jump=None, step=1, ns=ns, nc=nc, n_max_frames=n_max_frames)
    nj = data_batcher.nj
    visible_frame_total = [d.idxs["pv"] for d in data_batcher.datasets]
    from .session import Placeholder, TrainSession
    loss, total_loss, total_loss_visible, placeholders = dgp_loss(data_batcher, dgp_cfg)
    pipeline = _augment(dgp_cfg)                 # before the expensive setup: a broken augmentation stack fails here
    trainer = _make_trainer(data_batcher, init_weights, max_frames=1, device=local_rank)
    learning_rate = Placeholder("learning_rate")
    train_op = loss.graph.minimize(total_loss_visible, learning_rate)          # fitdgp.py:412-418
    sess = TrainSession(trainer, loss.graph)
    saver = weights_io.Saver(max_to_keep=dgp_cfg.max_to_keep)     # fitdgp.py:401, 696
    uploader = _FrameUploader(trainer.device) if PREFETCH_DEPTH > 0 else None
    nepoch = int(np.min([int(data_batcher.n_visible_frames_total * dgp_cfg.n_times_all_frames), maxiters]))
    table = np.array([(i, vv) for i, v in enumerate(visible_frame_total) for vv in v]).reshape(-1, 2)
    batch_ind_all = np.random.randint(0, table.shape[0], size=nepoch)
    n_sched = batch_ind_all.shape[0]
    maxiters = max(1, n_sched // world)          # data-parallel: W windows per optimiser step
    data_batcher.reset()
    print("Begin Training for {} iterations".format(maxiters))
    t_start = time.time()
    it = -1

    def make_batch(it):                          # host side of one iteration (runs one iteration ahead on the prefetch thread)
        dataset_i, frame_i = table[batch_ind_all[_dp_index(it, n_sched)]]
        d = data_batcher.datasets[dataset_i]
        (vis, hid, _, images, joint_loc, _, _, addn), _ = data_batcher.next_batch(0, dataset_i, np.array([frame_i]),
                                                                                  np.array([], dtype=int))
        all_frame = np.sort(list(vis) + list(hid))
        vis_within = [int(np.where(all_frame == i)[0][0]) for i in vis]
        if pipeline is not None and dgp_cfg.wt == 0 and len(vis_within) > 0:      # fitdgp.py:481-482
            from ..augment import data_aug
            images, joint_loc = data_aug(images, vis_within, joint_loc, pipeline, dgp_cfg)
        lmap, lmask = _locref_targets(joint_loc, len(all_frame), vis_within, d.nx_out, d.ny_out, nj, dgp_cfg)
        feed_dict = _feed(placeholders, images, joint_loc, lmap, lmask, addn, None, None, 0, d.nx_out, d.ny_out, learning_rate,
                          dgp_cfg.lr)
        if uploader is not None:
            feed_dict[placeholders["inputs"]] = uploader(images)
        return dataset_i, frame_i, feed_dict

    for it, (dataset_i, frame_i, feed_dict) in enumerate(_prefetched(make_batch, maxiters)):
        t0 = time.time()
        loss_eval, _ = sess.run([loss, train_op], feed_dict)
        if it % displayiters == 0 and it > 0:
            print("\nIteration {}/{}".format(it, maxiters))
            print("dataset_i: ", dataset_i, " visible_frame_batch_i: ", [frame_i], flush=True)
            print(" running time: ", time.time() - t0, "\n loss: ", loss_eval, flush=True)
        if (it % saveiters == 0) or (it + 1) == maxiters:
            _save(saver, trainer, dgp_cfg.snapshot_prefix, step, it, (it + 1) == maxiters)
    print("Finished training {} iterations\n".format(it), flush=True)
    print("\n\n TOTAL TIME ELAPSED: ", time.time() - t_start)
    return None


def fit_dgp(snapshot, dlcpath, batch_size=10, shuffle=1, step=2, saveiters=1000, displayiters=5, maxiters=200000, ns=10,
            nc=2048, n_max_frames=2000, gm2=0, gm3=0, nepoch=100, wt=0, aug=True, debug="", trainingsetindex=0,
            frame_sources=None):
    """Run DGP (fitdgp.py:549-845): batches of `batch_size` consecutive frames of the selected windows, at least
    one labeled frame per batch, loss = total_loss (visible + hidden CE, locref, spatial clique)."""
    rank, world, local_rank = _dp_setup()
    data_batcher, init_weights = _setup(snapshot, dlcpath, shuffle, trainingsetindex, frame_sources)
    dgp_cfg = data_batcher.dlc_config
    dgp_cfg.update(ws=1000, ws_max=1.2, wt=wt, wt_max=0, wn_visible=5, wn_hidden=3, gamma=1, gauss_len=1, lengthscale=1,
                   max_to_keep=5, batch_size=batch_size, n_times_all_frames=nepoch, lr=0.005, gm2=gm2, gm3=gm3, aug=aug)
    from .. import weights_io
    final = dgp_cfg.snapshot_prefix + "-step{}{}-final--0".format(step, debug)
    if weights_io.exists(final):
        print(final, "  exists! DGP has already been run.", flush=True)
        return None
    data_batcher.create_batches_from_resnet_output(0, ns_jump=None, step=1, ns=ns, nc=nc, n_max_frames=n_max_frames)
    nj = data_batcher.nj
    print("n_hidden_frames_total", data_batcher.n_frames_total - data_batcher.n_visible_frames_total, flush=True)
    print("n_visible_frames_total", data_batcher.n_visible_frames_total, flush=True)
    print("n_frames_total", data_batcher.n_frames_total, flush=True)
    visible_frame_total = [d.idxs["pv"] for d in data_batcher.datasets]
    hidden_frame_total = [d.idxs["ph"] for d in data_batcher.datasets]
    all_frame_total = [d.idxs["chunk"] for d in data_batcher.datasets]
    from .session import Placeholder, TrainSession
    loss, total_loss, total_loss_visible, placeholders = dgp_loss(data_batcher, dgp_cfg)
    pipeline = _augment(dgp_cfg)
    trainer = _make_trainer(data_batcher, init_weights, max_frames=batch_size + 1, device=local_rank)
    learning_rate = Placeholder("learning_rate")
    train_op = loss.graph.minimize(total_loss, learning_rate)                  # fitdgp.py:708-713
    sess = TrainSession(trainer, loss.graph)
    saver = weights_io.Saver(max_to_keep=dgp_cfg.max_to_keep)     # fitdgp.py:401, 696
    uploader = _FrameUploader(trainer.device) if PREFETCH_DEPTH > 0 else None
    batch_ind_all = gen_batch(visible_frame_total, hidden_frame_total, all_frame_total, dgp_cfg, maxiters)
    save_iters = max(int(saveiters / dgp_cfg.batch_size), 1)
    n_sched = len(batch_ind_all)
    maxiters = max(1, n_sched // world)
    data_batcher.reset()
    print("Begin Training for {} iterations".format(maxiters))
    t_start = time.time()
    it = -1

    def make_batch(it):                          # host side of one iteration (runs one iteration ahead on the prefetch thread)
        batch_ind = batch_ind_all[_dp_index(it, n_sched)]
        dataset_i = int(batch_ind[-1])
        d = data_batcher.datasets[dataset_i]
        all_frame_batch = batch_ind[:-1]
        visible_frame_i = visible_frame_total[dataset_i]
        all_frame_i = list(all_frame_total[dataset_i]) + list(hidden_frame_total[dataset_i])
        vis_b = np.sort(np.array([i for i in all_frame_batch if i in visible_frame_i], dtype=int))
        if len(vis_b) == 0 and len(visible_frame_i) > 0:              # guarantee one labeled frame (fitdgp.py:755-758)
            vis_b = np.array([visible_frame_i[randint(0, len(visible_frame_i) - 1)]])
        hid_b = np.sort(np.array([i for i in all_frame_batch if (i in all_frame_i) and (i not in visible_frame_i)],
                                 dtype=int))
        (vis, hid, _, images, joint_loc, wt_mask, _, addn), _ = data_batcher.next_batch(0, dataset_i, vis_b, hid_b)
        all_frame = np.sort(list(vis) + list(hid))
        vis_within = [int(np.where(all_frame == i)[0][0]) for i in vis]
        if pipeline is not None and dgp_cfg.wt == 0 and len(vis_within) > 0:      # fitdgp.py:778-779
            from ..augment import data_aug
            images, joint_loc = data_aug(images, vis_within, joint_loc, pipeline, dgp_cfg)
        lmap, lmask = _locref_targets(joint_loc, len(all_frame), vis_within, d.nx_out, d.ny_out, nj, dgp_cfg)
        vector_field = None
        if dgp_cfg.wt > 0:                                            # temporal clique: flow field from the host hook
            from .fitdgp_util import learn_wt
            vector_field = learn_wt(images)
        feed_dict = _feed(placeholders, images, joint_loc, lmap, lmask, addn, wt_mask, vector_field, dgp_cfg.wt, d.nx_out,
                          d.ny_out, learning_rate, dgp_cfg.lr)
        if uploader is not None:
            feed_dict[placeholders["inputs"]] = uploader(images)
        return dataset_i, vis_b, hid_b, feed_dict

    for it, (dataset_i, vis_b, hid_b, feed_dict) in enumerate(_prefetched(make_batch, maxiters)):
        t0 = time.time()
        loss_eval, _ = sess.run([loss, train_op], feed_dict)
        if it % displayiters == 0 and it > 0:
            print("\nIteration {}/{}".format(it, maxiters))
            print("dataset_i: ", dataset_i, flush=True)
            print("visible_frame_batch_i: ", vis_b, flush=True)
            print("hidden_frame_batch_i: ", hid_b, flush=True)
            print("\n running time: ", time.time() - t0, flush=True)
            print("\n loss: ", loss_eval, flush=True)
        if (it % save_iters == 0) or (it + 1) == maxiters:
            _save(saver, trainer, dgp_cfg.snapshot_prefix, step, it, (it + 1) == maxiters, debug)
    print("Finished {} iterations\n".format(it), flush=True)
    print("\n\n TOTAL TIME ELAPSED: ", time.time() - t_start)
    return None
