"""Random submit patterns through DGPPipeline (1-3 engines on as many HIP streams: batch sizes that change from submit to submit, joins at random
points, output buffers reused as soon as their event has completed) against ONE engine calibrated on the same batch: every packed trajectory must
be bit-identical, whichever engine computed it.  Usage: python scripts/fuzz_pipeline.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgraphpose_amd import engine
from deepgraphpose_amd.synthetic import make_frames, make_weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(n):
    nj, ns, B = int(rng.integers(1, 6)), int(rng.integers(1, 4)), int(rng.integers(2, 9))
    h, w = int(rng.integers(48, 130)), int(rng.integers(48, 150))
    wts = make_weights(50, nj, False, seed=int(rng.integers(0, 1000)), head_std=0.05)
    pool = torch.from_numpy(make_frames(4 * B, h, w, nj, seed=int(rng.integers(0, 1000)))).cuda()
    cal = pool[:B].contiguous()
    ref = engine.DGPNet(50, nj, h, w, max_batch=B); ref.load_weights(wts)
    scratch = torch.zeros((B, nj, 5), device="cuda")
    ref.infer_packed(cal, scratch)                                     # calibrates on `cal`
    pipe = engine.DGPPipeline(50, nj, h, w, max_batch=B, n_streams=ns); pipe.load_weights(wts)
    pipe.calibrate(cal)
    nsub = int(rng.integers(5, 25))
    outs, evs, wants = [], [], []
    bufs = [torch.zeros((B, nj, 5), device="cuda") for _ in range(ns + 1)]
    pending = {}
    ok = True
    for k in range(nsub):
        b = int(rng.integers(1, B + 1)); off = int(rng.integers(0, 4 * B - b + 1))
        fr = pool[off:off + b].contiguous()
        slot = k % len(bufs)
        if slot in pending:                                             # reuse a buffer only after its forward has completed: check it first
            ev, want, got_view, bb = pending.pop(slot)
            ev.synchronize()
            ok = ok and torch.equal(got_view[:bb], want)
        want = torch.zeros((b, nj, 5), device="cuda"); ref.infer_packed(fr, want)
        torch.cuda.synchronize()                                        # (the reference engine runs on the caller's stream)
        ev = pipe.submit(fr, bufs[slot][:b])
        pending[slot] = (ev, want.clone(), bufs[slot], b)
        if rng.integers(0, 4) == 0:
            pipe.join(); torch.cuda.synchronize()
    pipe.join(); torch.cuda.synchronize()
    for slot, (ev, want, got_view, bb) in pending.items():
        ok = ok and torch.equal(got_view[:bb], want)
    ok = ok and not pipe.range_status()[0] and not ref.range_status()[0]
    bad += not ok
    print("%s %d engines, nj %d, %3d x %3d, max batch %d, %d submits" % ("ok  " if ok else "BAD ", ns, nj, h, w, B, nsub), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
