"""Soak: N forwards of the bench workload (ResNet-50, 640x480, batch 32, both heads) on the same frames; every scoremap / locref / keypoint
output must equal the first run bit for bit (a race in a loader protocol would show up as run-to-run differences).  Usage: python scripts/soak_determinism.py [N] [parity|f16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepgraphpose_amd import engine, synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
net = engine.DGPNet(50, 4, 480, 640, max_batch=32, tier=(sys.argv[2] if len(sys.argv) > 2 else "parity"))
net.load_weights(synthetic.make_weights(50, 4, True, seed=0))
f = torch.from_numpy(synthetic.make_frames(32, 480, 640, 4, seed=1)).cuda()
def run():
    out = net.forward(f)
    out = out if isinstance(out, (tuple, list)) else (out,)
    mu, conf, idx = net.infer(f, check_range=False)
    return [o.clone() for o in out] + [mu.clone(), conf.clone(), idx.clone()]
ref = run()
bad = 0
for i in range(N):
    cur = run()
    for k, (a, b) in enumerate(zip(ref, cur)):
        if not torch.equal(a, b):
            bad += 1
            print("run %d output %d differs: max |d| %.3e" % (i, k, float((a.float() - b.float()).abs().max())), flush=True)
    if bad > 5:
        break
assert not net.range_status()[0]
print("soak: %d runs, %d differing outputs" % (N, bad))
sys.exit(1 if bad else 0)
