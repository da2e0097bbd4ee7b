#!/usr/bin/env python3
"""Where does a one-node RCCL start-up spend its time?  Run under torch.distributed.run; prints per-phase seconds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
import torch, torch.distributed as dist
from deepgraphpose_amd import dist as ddist
t1 = time.perf_counter()
rank, lr, world = ddist.init_from_env("nccl")
t2 = time.perf_counter()
dist.barrier(device_ids=[lr]); torch.cuda.synchronize()
t3 = time.perf_counter()
x = torch.ones(1024, device="cuda")
out = torch.empty(1024 * world, device="cuda")
dist.all_gather_into_tensor(out, x); torch.cuda.synchronize()
t4 = time.perf_counter()
dist.barrier(device_ids=[lr]); torch.cuda.synchronize()
t5 = time.perf_counter()
print("RCCL probe: import %.1f s, init_process_group %.1f s, first barrier %.1f s, first all_gather %.1f s, second barrier %.3f s | env %s"
      % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, {k: v for k, v in os.environ.items() if k.startswith("NCCL") or k.startswith("RCCL")}))
dist.destroy_process_group()
