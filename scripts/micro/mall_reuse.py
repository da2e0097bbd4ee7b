"""Does a tensor written by one launch come back from the Infinity Cache (256 MB, memory side) when the next launch reads it?
Reads of an S-MB buffer right after it was written ("warm") against reads after 2 GB of other traffic ("flushed"), HIP-event timed.
Usage: python scripts/micro/mall_reuse.py"""
import torch
dev = torch.device("cuda")
big = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for mb in (39, 78, 157, 236, 314, 628, 1256):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, dtype=torch.float32, device=dev)
    y = torch.empty(n, dtype=torch.float32, device=dev)
    res = {}
    for mode in ("warm", "flushed"):
        ts = []
        for _ in range(7):
            x.fill_(1.0)                      # the producer: writes S MB
            if mode == "flushed":
                big.fill_(0)                  # 2 GB of other traffic in between
            torch.cuda.synchronize()
            e0.record()
            y.copy_(x)                        # the consumer: reads S MB (and writes S MB)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        res[mode] = ts[len(ts) // 2]
    print("%5d MB  copy after write: warm %.3f ms (%.2f TB/s read+write)   flushed %.3f ms (%.2f TB/s)   warm/flushed %.2f"
          % (mb, res["warm"], 2 * mb * 1.048576e-3 / res["warm"], res["flushed"], 2 * mb * 1.048576e-3 / res["flushed"], res["warm"] / res["flushed"]), flush=True)
    del x, y
