"""One H2 conv layer a few times (for rocprofv3 kernel traces / PMC passes of a single shape).
Usage: python scripts/h2_conv_once.py N H W Cin Cout k stride rate [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgraphpose_amd import engine
N, H, W, Cin, Cout, k, stride, rate = (int(v) for v in sys.argv[1:9])
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 5
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.relu(torch.randn((N, H, W, Cin), device="cuda", generator=g)) * 3.0
w = (np.random.default_rng(1).standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
pad = ((k - 1) * rate) // 2
xe = engine.h2_exp_for(float(x.abs().max()))
xh = engine.f32_to_h2(x, xe)
for _ in range(reps):
    y, r = engine.conv2d_h2(xh, xe, w, stride=stride, rate=rate, pad_t=pad, pad_l=pad, out_hw=(H, W), relu=True, y_is_h2=True, y_exp=xe)
torch.cuda.synchronize()
print("done", float(r.max()))
