"""Per-layer timing of wgrad_f32<1> (the 64-channel layers of the training step at 11 frames of 640x480) against each layer's HBM and fp32-MFMA floors."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from deepgraphpose_amd import _lib
from deepgraphpose_amd.engine import _conv_desc, _ptr, _stream
CASES = [("stem 7x7/2 3(4)->64", 11, 480, 640, 4, 64, 7, 2, 1, (240, 320)),
         ("block1 conv1 64->64", 11, 120, 160, 64, 64, 1, 1, 1, (120, 160)),
         ("block1 conv1 256->64", 11, 120, 160, 256, 64, 1, 1, 1, (120, 160)),
         ("block1 conv2 64->64 3x3", 11, 120, 160, 64, 64, 3, 1, 1, (120, 160)),
         ("block1 conv2 3x3 /2", 11, 120, 160, 64, 64, 3, 2, 1, (60, 80)),
         ("block1 conv3 64->256", 11, 120, 160, 64, 256, 1, 1, 1, (120, 160)),
         ("block1 shortcut 64->256", 11, 120, 160, 64, 256, 1, 1, 1, (120, 160))]
lib = _lib.load(); dev = torch.device("cuda")
for name, N, H, W, Cin, Cout, k, stride, rate, ohw in CASES:
    pad = (k - 1) * rate // 2
    x = torch.relu(torch.randn((N, H, W, Cin), device=dev)); dy = torch.randn((N, ohw[0], ohw[1], Cout), device=dev) * 1e-3
    d = _conv_desc(x.shape, (k, k, Cin, Cout), stride, rate, pad, pad, ohw)
    dw = torch.empty((k, k, Cin, Cout), device=dev); cs = torch.empty(2 * Cout, device=dev)
    rng = torch.zeros((2, 256), device=dev)
    ranged = "--ranged" in sys.argv          # hand the operand ranges over, as the training step does (16-bit tiles may be chosen)
    if ranged:
        lib.dgp_tensor_absmax(_ptr(x), x.numel(), _ptr(rng[0]), _stream(dev)); lib.dgp_tensor_absmax(_ptr(dy), dy.numel(), _ptr(rng[1]), _stream(dev))
    run = lambda: lib.dgp_conv2d_wgrad(C.byref(d), _ptr(x), _ptr(dy), _ptr(rng[0]) if ranged else None, _ptr(rng[1]) if ranged else None,
                                       _ptr(dw), _ptr(cs), _stream(dev))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gf = 2.0 * N * ohw[0] * ohw[1] * Cin * Cout * k * k / 1e9
    mb = (x.numel() + dy.numel()) * 4 / 1e6
    print("%-26s %6.1f GFLOP %6.1f MB  %7.1f us   floors: HBM (5 TB/s) %5.1f us, fp32 MFMA %5.1f us, 16-bit x3 %5.1f us" % (name, gf, mb, us, mb / 5.0, gf / 157.3 * 1e3, gf * 3 / 2500 * 1e3), flush=True)
