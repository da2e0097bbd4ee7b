"""Per-layer timing of the weight-gradient kernels at the config-4 shapes: wgrad_h3p (fp32 operands, split in the kernel) against
wgrad_dma (fp16 high / low copies by LDS-DMA; the copies are made outside the timed region as the producers' epilogues make them)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from deepgraphpose_amd import _lib, engine
from deepgraphpose_amd.engine import _conv_desc, _ptr, _stream, ABSMAX_SLOTS

CASES = [
    ("block4 conv2", 11, 30, 40, 512, 512, 3, 1, 2),
    ("block4 conv1", 11, 30, 40, 2048, 512, 1, 1, 1),
    ("block4 conv3", 11, 30, 40, 512, 2048, 1, 1, 1),
    ("block3 conv2", 11, 30, 40, 256, 256, 3, 1, 1),
    ("block3 conv1", 11, 30, 40, 1024, 256, 1, 1, 1),
    ("block3 conv3", 11, 30, 40, 256, 1024, 1, 1, 1),
    ("block2 conv2", 11, 60, 80, 128, 128, 3, 1, 1),
    ("block2 conv1", 11, 60, 80, 512, 128, 1, 1, 1),
    ("block2 conv3", 11, 60, 80, 128, 512, 1, 1, 1),
]


def main():
    lib = _lib.load()
    dev = torch.device("cuda")
    reps = 20
    for name, N, H, W, Cin, Cout, k, stride, rate in CASES:
        pad = (k - 1) * rate // 2
        x = torch.relu(torch.randn((N, H, W, Cin), device=dev))
        dy = torch.randn((N, H, W, Cout), device=dev) * 1e-3
        d = _conv_desc(x.shape, (k, k, Cin, Cout), stride, rate, pad, pad, (H, W))
        dw = torch.empty((k, k, Cin, Cout), device=dev)
        cs = torch.empty(2 * Cout, device=dev)
        rng = torch.zeros((2, ABSMAX_SLOTS), device=dev)
        lib.dgp_tensor_absmax(_ptr(x), x.numel(), _ptr(rng[0]), _stream(dev))
        lib.dgp_tensor_absmax(_ptr(dy), dy.numel(), _ptr(rng[1]), _stream(dev))
        scratch = torch.empty(x.numel() + dy.numel(), device=dev)
        out = {}
        for which in ("h3p", "dma+copies"):
            def run():
                if which == "h3p":
                    lib.dgp_conv2d_wgrad(C.byref(d), _ptr(x), _ptr(dy), _ptr(rng[0]), _ptr(rng[1]), _ptr(dw), _ptr(cs), _stream(dev))
                else:
                    lib.dgp_conv2d_wgrad_shadow(C.byref(d), _ptr(x), _ptr(dy), _ptr(rng[0]), _ptr(rng[1]), _ptr(rng[0]), _ptr(rng[1]),
                                                _ptr(scratch), _ptr(dw), _ptr(cs), _stream(dev))
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            out[which] = e0.elapsed_time(e1) / reps * 1e3
        gf = 2.0 * N * H * W * Cin * Cout * k * k / 1e9
        print(f"{name:14s} {gf:7.1f} GFLOP  h3p(+memset) {out['h3p']:7.1f} us  dma(+memset+2 copies) {out['dma+copies']:7.1f} us", flush=True)


if __name__ == "__main__":
    sys.exit(main())
