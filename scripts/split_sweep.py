#!/usr/bin/env python3
"""Split-bf16 conv kernels vs the fp32-MFMA kernels: TFLOP/s (fp32-equivalent FLOPs) and error vs an fp64 reference
on sampled outputs, per ResNet-50 640x480 batch-32 layer shape.  Usage: split_sweep.py [name filter ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from deepgraphpose_amd import engine, _lib
from scripts.conv_sweep_layers import LAYERS

TILES = [(int(t.split(":")[0]), t.split(":")[1]) for t in os.environ.get("SWEEP_TILES", "4:W8,10:S6k16,12:S6k16w8,13:H3k16,14:H3k16w8,9:S6n64,15:H3n64").split(",")]
only = sys.argv[1:]
rng = np.random.default_rng(0)
lib = _lib.load()
for name, N, H, W, Cin, Cout, k, s, r, pad in LAYERS:
    if only and not any(o in name for o in only):
        continue
    N = N * int(os.environ.get("SWEEP_BATCH_MULT", "1")) // int(os.environ.get("SWEEP_BATCH_DIV", "1"))
    x = torch.randn((N, H, W, Cin), device="cuda") * torch.rand((1, 1, 1, Cin), device="cuda") * 3
    x = torch.relu(x)                                     # post-ReLU-like activations
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    flops = 2.0 * N * H * W * k * k * Cin * Cout
    wp = torch.from_numpy(engine.pack_conv_weights(w)).cuda()
    d = _lib.DgpConvDesc(N, H, W, Cin, Cout, k, k, s, r, pad, pad, H, W, 0, 0, 0, 0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng_dev = torch.zeros(3 * 256, device="cuda")
    _lib.check(lib.dgp_tensor_absmax(C.c_void_p(x.data_ptr()), x.numel(), C.c_void_p(rng_dev.data_ptr()), st))
    _lib.check(lib.dgp_tensor_absmax(C.c_void_p(wp.data_ptr()), wp.numel(), C.c_void_p(rng_dev.data_ptr() + 1024), st))
    # fp64 reference on 64 sampled pixels (all channels)
    pix = rng.integers(0, N * H * W, 64)
    xw = torch.nn.functional.pad(x.double(), (0, 0, pad, pad, pad, pad))
    wt = torch.from_numpy(w).double().cuda()
    ref = torch.zeros((64, Cout), dtype=torch.float64, device="cuda")
    for i, pm in enumerate(pix):
        n, rem = divmod(int(pm), H * W); ho, wo = divmod(rem, W)
        patch = xw[n, ho:ho + r * (k - 1) + 1:r, wo:wo + r * (k - 1) + 1:r, :]        # [k,k,Cin]
        ref[i] = torch.einsum("abc,abco->o", patch, wt)
    out = []
    for tile, label in TILES:
        if tile in (0, 4, 5, 7, 8, 10, 11, 12, 13, 14, 16, 17) and Cout % 128:
            out.append("%s   -   " % label); continue
        os.environ["DGP_FORCE_TILE"] = str(tile)
        y = torch.empty((N, H, W, Cout), device="cuda")
        def run():
            _lib.check(lib.dgp_conv2d_ranged(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), None, None, None,
                                             C.c_void_p(y.data_ptr()), C.c_void_p(rng_dev.data_ptr()),
                                             C.c_void_p(rng_dev.data_ptr() + 1024), C.c_void_p(rng_dev.data_ptr() + 2048), st))
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        got = y.view(-1, Cout)[torch.from_numpy(pix).cuda()].double()
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        out.append("%s %6.1f TF err %.1e" % (label, flops / ms / 1e9, err))
    print("%-22s | %s" % (name, " | ".join(out)), flush=True)
