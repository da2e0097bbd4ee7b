"""Degenerate arguments through the Python wrappers of the C-ABI: empty batches, zero-sized maps, batches beyond max_batch, 1 x 1 frames, frames
smaller than the stride.  Every call must either return a well-defined (empty) result or raise DgpError / ValueError -- never crash or hang.
Usage: python scripts/probe_edges.py"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgraphpose_amd import engine, _lib
from deepgraphpose_amd.synthetic import make_frames, make_weights
bad = 0
def probe(name, fn, expect=None):
    global bad
    try:
        r = fn()
        torch.cuda.synchronize()
        desc = "returned " + (", ".join(str(tuple(x.shape)) for x in r) if isinstance(r, (tuple, list)) else (str(tuple(r.shape)) if hasattr(r, "shape") else repr(r)[:60]))
        ok = expect in (None, "ok")
    except (_lib.DgpError, ValueError, AssertionError, RuntimeError) as e:
        desc = "%s: %s" % (type(e).__name__, str(e)[:110]); ok = expect in (None, "error")
    except Exception as e:      # noqa: BLE001
        desc = "UNEXPECTED %s: %s" % (type(e).__name__, str(e)[:110]); ok = False
    bad += not ok
    print("%s %-58s %s" % ("ok  " if ok else "BAD ", name, desc), flush=True)

dev = "cuda"
z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
probe("soft_argmax on an empty batch [0, 8, 8, 3]", lambda: engine.soft_argmax(z(0, 8, 8, 3)), "ok")
probe("soft_argmax with zero joints [2, 8, 8, 0]", lambda: engine.soft_argmax(z(2, 8, 8, 0)), "ok")
probe("soft_argmax on a 0 x 8 map", lambda: engine.soft_argmax(z(1, 0, 8, 2)), "error")
probe("soft_argmax on a 1 x 1 map", lambda: engine.soft_argmax(z(1, 1, 1, 2)), "ok")
probe("soft_argmax with gauss_len 0", lambda: engine.soft_argmax(z(1, 6, 6, 2), 1.0, 0))
probe("soft_argmax with gauss_len 40 on a 6 x 6 map", lambda: engine.soft_argmax(z(1, 6, 6, 2), 1.0, 40))
probe("soft_argmax with gamma 0", lambda: engine.soft_argmax(z(1, 6, 6, 2), 0.0, 1), "ok")
probe("soft_argmax with NaN scores", lambda: engine.soft_argmax(z(1, 6, 6, 2) + float("nan"), 1.0, 1))
probe("hard_argmax on an empty batch", lambda: engine.hard_argmax(z(0, 8, 8, 3)), "ok")
probe("pmap_threshold on an empty batch", lambda: engine.pmap_threshold(z(0, 8, 8, 3), 0.5), "ok")
probe("motion_energy of zero frames", lambda: engine.motion_energy(torch.zeros((0, 8, 8, 3), dtype=torch.uint8, device=dev)), "ok")
probe("motion_energy of one frame", lambda: engine.motion_energy(torch.zeros((1, 8, 8, 3), dtype=torch.uint8, device=dev)), "ok")
probe("maxpool of a 1 x 1 map", lambda: engine.maxpool_3x3s2_same(z(1, 1, 1, 8)), "ok")
probe("maxpool of an empty batch", lambda: engine.maxpool_3x3s2_same(z(0, 4, 4, 8)))
probe("conv2d on an empty batch", lambda: engine.conv2d(z(0, 8, 8, 32), np.zeros((1, 1, 32, 64), np.float32)))
probe("conv2d with Cin = 3 (not a multiple of 4)", lambda: engine.conv2d(z(1, 8, 8, 3), np.zeros((1, 1, 3, 64), np.float32)), "error")
probe("conv2d with a kernel larger than the padded map", lambda: engine.conv2d(z(1, 2, 2, 32), np.zeros((3, 3, 32, 64), np.float32)))
wts = make_weights(50, 3, False, seed=1, head_std=0.05)
net = engine.DGPNet(50, 3, 64, 96, max_batch=2); net.load_weights(wts)
u8 = lambda *s: torch.zeros(s, dtype=torch.uint8, device=dev)
probe("infer on an empty batch", lambda: net.infer(u8(0, 64, 96, 3)), "ok")
probe("forward on an empty batch", lambda: net.forward(u8(0, 64, 96, 3)), "ok")
probe("infer on a batch beyond max_batch", lambda: net.infer(u8(3, 64, 96, 3)), "error")
probe("infer on frames of another size than planned", lambda: net.infer(u8(1, 60, 96, 3)), "error")
probe("infer on float frames", lambda: net.infer(z(1, 64, 96, 3)), "error")
probe("infer after the weights of ANOTHER head size", lambda: net.load_weights(make_weights(50, 5, False, seed=1)), "error")
probe("set_input_size to 1 x 1", lambda: (net.set_input_size(1, 1), net.infer(u8(1, 1, 1, 3)))[1])
probe("set_input_size to 7 x 9 (smaller than the stride)", lambda: (net.set_input_size(7, 9), net.infer(u8(2, 7, 9, 3)))[1])
probe("set_input_size to 0 x 10", lambda: net.set_input_size(0, 10), "error")
probe("back to 64 x 96 and infer", lambda: (net.set_input_size(64, 96), net.infer(torch.from_numpy(make_frames(2, 64, 96, 3, seed=2)).cuda()))[1], "ok")
probe("DGPNet with 0 joints", lambda: engine.DGPNet(50, 0, 64, 96, max_batch=1), "error")
probe("DGPNet with depth 34", lambda: engine.DGPNet(34, 3, 64, 96, max_batch=1), "error")
probe("DGPNet with max_batch 0", lambda: engine.DGPNet(50, 3, 64, 96, max_batch=0), "error")
print("failures:", bad)
sys.exit(1 if bad else 0)
