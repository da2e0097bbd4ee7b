"""Random uint8 clips through dgp_motion_energy against the oracle's restatement of calculate_motion_energy (bit-exact contract): frame sizes
from 1 byte to odd, unaligned sizes, 1-40 frames, chunked with `prev`.  Usage: python scripts/fuzz_motion.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgraphpose_amd import engine
from oracle import dgp_oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = 0
for it in range(n):
    T = int(rng.integers(1, 41))
    H, W = (int(rng.integers(1, 4)), int(rng.integers(1, 4))) if rng.integers(0, 6) == 0 else (int(rng.integers(1, 300)), int(rng.integers(1, 400)))
    kind = rng.integers(0, 4)
    clip = rng.integers(0, 256, (T, H, W, 3), dtype=np.uint8)
    if kind == 1:
        clip[:] = clip[0]                                     # static clip: all zeros
    elif kind == 2:
        clip = (np.arange(T, dtype=np.int64)[:, None, None, None] * 37 + clip[0].astype(np.int64)).astype(np.uint8)      # every difference wraps the same way
    want = O.motion_energy(clip)
    got = engine.motion_energy(torch.from_numpy(clip).cuda())
    ok = np.array_equal(got, want)
    if T > 2:                                                 # the same clip in two chunks, the second with `prev`
        k = int(rng.integers(1, T))
        a = engine.motion_energy(torch.from_numpy(clip[:k]).cuda())
        b = engine.motion_energy(torch.from_numpy(clip[k:]).cuda(), prev=torch.from_numpy(clip[k - 1]).cuda())
        ok = ok and np.array_equal(np.concatenate([a, b]), want)
    fails += 0 if ok else 1
    if not ok or it % 20 == 0:
        print("%s T %2d %3d x %3d kind %d" % ("ok  " if ok else "FAIL", T, H, W, kind), flush=True)
print("failures: %d" % fails)
sys.exit(1 if fails else 0)
