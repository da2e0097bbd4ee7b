"""Frame-sharded multi-GPU inference: one process per GPU, RCCL all-gather over xGMI.

Frames of a video are independent (estimate_pose has no cross-frame state,
DGP/models/eval.py:306-345), so rank r takes the contiguous block
[r*ceil(T/W), (r+1)*ceil(T/W)) and the only exchange is ONE all-gather per video (or per
chunk of frames) of the per-frame keypoints -- 20 bytes per (frame, joint): (row, col, conf)
fp32 + (iy, ix) int32 bit-cast into the same 4-byte lanes so a single collective moves both.
The payload is tiny (latency-bound), which is why it is done once per chunk, never per batch.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of frame indices owned by `rank` (last shards may be short or empty)."""
    per = -(-n_frames // world) if n_frames > 0 else 0
    lo = min(rank * per, n_frames)
    hi = min(lo + per, n_frames)
    return lo, hi


def pack_keypoints(mu: torch.Tensor, conf: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """[T,nj,2] f32, [T,nj] f32, [T,nj,2] i32 -> [T,nj,5] f32 (idx bit-cast)."""
    return torch.cat([mu, conf.unsqueeze(-1), idx.contiguous().view(torch.float32)], dim=-1).contiguous()


def unpack_keypoints(buf: torch.Tensor):
    mu = buf[..., 0:2].contiguous()
    conf = buf[..., 2].contiguous()
    idx = buf[..., 3:5].contiguous().view(torch.int32)
    return mu, conf, idx


def gather_trajectory(local: torch.Tensor, n_frames: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """All-gather per-rank packed keypoints [T_r, nj, 5] into the frame-ordered [T, nj, 5]
    on every rank.  Shards are padded to ceil(T/W) so one fixed-size all-gather suffices."""
    if not dist.is_available() or not dist.is_initialized():
        return local[:n_frames]
    world = dist.get_world_size(group)
    per = -(-n_frames // world)
    nj, k = local.shape[1], local.shape[2]
    send = local
    if local.shape[0] != per:
        send = torch.zeros((per, nj, k), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    if dist.get_backend(group) == "gloo" and send.is_cuda:       # gloo has no device all-gather: stage through the host (tests; CPU-only groups)
        parts = [torch.empty((per, nj, k), dtype=local.dtype) for _ in range(world)]
        dist.all_gather(parts, send.contiguous().cpu(), group=group)
        return torch.cat(parts, 0)[:n_frames].to(local.device)
    out = torch.empty((world * per, nj, k), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, send.contiguous(), group=group)
    return out[:n_frames]


def any_rank(flag: bool, device=None, group: Optional[dist.ProcessGroup] = None) -> bool:
    """True on every rank when `flag` is true on at least one (one all-reduce(MAX) of a single int)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bool(flag)
    on_host = dist.get_backend(group) == "gloo" or device is None
    t = torch.tensor([int(bool(flag))], dtype=torch.int32, device="cpu" if on_host else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(t.item())


def from_rank0(flag: bool, group: Optional[dist.ProcessGroup] = None) -> bool:
    """Rank 0's value of `flag` on every rank (a decision all ranks must take together, e.g. "the labels already exist")."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bool(flag)
    box = [bool(flag)]
    dist.broadcast_object_list(box, src=0, group=group)
    return bool(box[0])


def average_gradients(flat_grads: torch.Tensor, group: Optional[dist.ProcessGroup] = None,
                      bucket_floats: int = 16 * 1024 * 1024) -> torch.Tensor:
    """Data-parallel training: mean of the flat gradient buffer over the ranks, in place.

    The trainer keeps every trainable tensor in ONE flat fp32 buffer (23.6 M floats = 94 MB for ResNet-50), so the
    exchange is a handful of large ring all-reduces over RCCL/xGMI (64 MB buckets: big enough to run at link rate,
    small enough that the tail of bucket k overlaps the reduction of bucket k+1) instead of ~160 per-tensor calls.
    Each rank then applies the same clip + momentum update, so the replicas stay bit-identical.  The reference
    trains on a single GPU; with W ranks each step consumes W windows of frames (one per rank)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return flat_grads
    world = dist.get_world_size(group)
    flat = flat_grads.view(-1)
    handles = []
    for lo in range(0, flat.numel(), bucket_floats):
        handles.append(dist.all_reduce(flat[lo:lo + bucket_floats], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for h in handles:
        h.wait()
    flat.mul_(1.0 / world)
    return flat_grads


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from torchrun's env; initialises the process group when world > 1.
    backend 'nccl' is RCCL on ROCm; 'gloo' is used by the CPU tests."""
    import os
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ       # under torchrun, even with 1 rank
    if (world > 1 or launched) and not dist.is_initialized():
        if backend is None or os.environ.get("DGP_DIST_BACKEND"):      # DGP_DIST_BACKEND=gloo: several ranks on ONE GPU (tests)
            backend = os.environ.get("DGP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def parse_cpulist(text: str):
    """'0-3,8,10-11' (sysfs cpulist syntax) -> sorted list of CPU numbers."""
    cpus = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            cpus.update(range(int(lo), int(hi) + 1))
        else:
            cpus.add(int(part))
    return sorted(cpus)


def device_identity(local_rank: int) -> dict:
    """What a reader needs to verify which GPU a rank really ran on: torch's device index, the PCI address, the device uuid (when
    torch exposes it) and the NUMA node sysfs reports for that PCI function (-1: unknown / single node)."""
    pr = torch.cuda.get_device_properties(local_rank)
    bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
    node = -1
    try:
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
            node = int(f.read().strip())
    except (OSError, ValueError):
        pass
    return {"index": int(local_rank), "name": pr.name, "pci": bdf, "uuid": str(getattr(pr, "uuid", "")), "numa_node": node}


def bind_to_gpu_numa_node(local_rank: int) -> dict:
    """Pin this process (and every thread it starts afterwards: decode / staging / prefetch threads inherit the mask) to the CPUs of the
    NUMA node its GPU hangs off, so that with 8 ranks on one host the pinned staging buffers and the H2D copies of a rank stay on its
    GPU's side of the machine.  No-op when sysfs has no answer (single node, containers without /sys) or DGP_NUMA_BIND=0.
    -> {'numa_node', 'cpus_bound'} for the bench line."""
    import os
    ident = device_identity(local_rank)
    out = {"numa_node": ident["numa_node"], "cpus_bound": None}
    if os.environ.get("DGP_NUMA_BIND", "1") == "0" or ident["numa_node"] < 0 or not hasattr(os, "sched_setaffinity"):
        return out
    try:
        with open("/sys/devices/system/node/node%d/cpulist" % ident["numa_node"]) as f:
            cpus = set(parse_cpulist(f.read()))
        allowed = cpus & set(os.sched_getaffinity(0))      # (never widen a mask a launcher / cgroup already narrowed)
        if allowed:
            os.sched_setaffinity(0, allowed)
            out["cpus_bound"] = len(allowed)
    except (OSError, ValueError):
        pass
    return out
