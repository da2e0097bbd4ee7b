#!/usr/bin/env python3
"""Host-side cost of one fit_dgp iteration: the synthetic test project (64x96 frames: the GPU step is ~2 ms) under cProfile."""
import os, sys, time, tempfile, cProfile, pstats, io, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _project import make_project
from deepgraphpose_amd.models.fitdgp import fit_dgp
import pathlib
tmp = pathlib.Path(tempfile.mkdtemp())
hw = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 96)
proj, frames, wts = make_project(tmp, hw=hw)
np.random.seed(0); random.seed(0)
iters = 40
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
fit_dgp("snapshot-step0-final--0", proj, batch_size=10, shuffle=1, step=2, maxiters=iters, displayiters=1000, saveiters=100000, gm2=1, gm3=3,
        aug=True, n_max_frames=40, ns=10)
pr.disable()
dt = time.perf_counter() - t0
print("fit_dgp %dx%d: %.1f ms per iteration over %d iterations (incl. setup)" % (hw[0], hw[1], dt / iters * 1e3, iters))
sio = io.StringIO()
pstats.Stats(pr, stream=sio).sort_stats("cumulative").print_stats(28)
print("\n".join(l[:150] for l in sio.getvalue().splitlines()[:60]))
