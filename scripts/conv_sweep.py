#!/usr/bin/env python3
"""Time single conv layers of the ResNet-50 640x480 batch-32 plan under each tile config."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgraphpose_amd import engine

from scripts.conv_sweep_layers import LAYERS
BIG = int(os.environ.get("SWEEP_BATCH_MULT", "1"))
only = sys.argv[1:] 
rng = np.random.default_rng(0)
for name, N, H, W, Cin, Cout, k, s, r, pad in LAYERS:
    if only and not any(o in name for o in only):
        continue
    N = N * BIG
    x = torch.randn((N, H, W, Cin), device="cuda")
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    flops = 2.0 * N * H * W * k * k * Cin * Cout
    res = []
    for tile in (0, 1, 2, 4, 5, 6):
        if tile in (0, 4, 5) and Cout % 128: 
            res.append("   -  "); continue
        os.environ["DGP_FORCE_TILE"] = str(tile)
        y = engine.conv2d(x, w, stride=s, rate=r, pad_t=pad, pad_l=pad, out_hw=(H, W), relu=True)
        # time launches only: re-pack happens inside conv2d, so call the C-ABI directly in a loop
        import ctypes as C
        from deepgraphpose_amd import _lib
        lib = _lib.load()
        wp = torch.from_numpy(engine.pack_conv_weights(w)).cuda()
        d = _lib.DgpConvDesc(N, H, W, Cin, Cout, k, k, s, r, pad, pad, H, W, 1, 0, 0, 0)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        def run():
            _lib.check(lib.dgp_conv2d(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), None, None, None,
                                      C.c_void_p(y.data_ptr()), st))
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append("%6.1f" % (flops / ms / 1e9))
    print("%-22s GF %7.1f | TF 128x128 %s | 128x64 %s | 64x64 %s | 128x128/8w %s | LS128x128 %s | LS128x64 %s" % (name, flops / 1e9, res[0], res[1], res[2], res[3], res[4], res[5]), flush=True)
