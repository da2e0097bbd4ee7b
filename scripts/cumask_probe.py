#!/usr/bin/env python3
"""Probe: the two engines' streams with COMPLEMENTARY CU masks (hipExtStreamCreateWithCUMask: each stream owns half of the 256 CUs) against the
plain two-stream arrangement -- does a static half-chip partition (finer tile quantisation per launch, private L2s) beat letting the two
streams' workgroups mix?  python scripts/cumask_probe.py [parity|f16] [pattern: half|even]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepgraphpose_amd import engine, synthetic
tier = sys.argv[1] if len(sys.argv) > 1 else "parity"
pattern = sys.argv[2] if len(sys.argv) > 2 else "half"
H, W, NJ, B = 480, 640, 4, 32
hip = ctypes.CDLL("libamdhip64.so")
torch.cuda.init(); torch.zeros(1, device="cuda")
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits(32 * w + b)) for w in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)
wts = synthetic.make_weights(50, NJ, False, seed=0, head_std=0.05)
frames = torch.from_numpy(synthetic.make_frames(B, H, W, NJ, seed=100)).cuda()
nets = [engine.DGPNet(50, NJ, H, W, max_batch=B, tier=tier) for _ in range(2)]
outs = [torch.zeros((B, NJ, 5), device="cuda") for _ in range(2)]
for n in nets:
    n.load_weights(wts)
def run(streams, K):
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            for _ in range(3):
                nets[i].infer_packed(frames, outs[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        with torch.cuda.stream(streams[i % 2]):
            nets[i % 2].infer_packed(frames, outs[i % 2])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K
plain = [torch.cuda.Stream(), torch.cuda.Stream()]
if pattern == "half":
    masked = [masked_stream(lambda c: c < 128), masked_stream(lambda c: c >= 128)]
else:
    masked = [masked_stream(lambda c: c % 2 == 0), masked_stream(lambda c: c % 2 == 1)]
ref = None
for rep in range(2):
    for name, st in (("plain", plain), ("masked(%s)" % pattern, masked)):
        dt = run(st, 200)
        print("tier %s %-14s %.3f ms per step, %.0f frames/s" % (tier, name, dt * 1e3, B / dt), flush=True)
one = run([masked[0], masked[0]], 60)
print("tier %s one masked stream alone (half the chip): %.3f ms per step, %.0f frames/s" % (tier, one * 1e3, B / one))
