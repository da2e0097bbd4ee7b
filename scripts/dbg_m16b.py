import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DGP_CONV2D_CELLS"] = "1"; os.environ["DGP_FORCE_TILE"] = "16"
import ctypes as C
import numpy as np, torch
from deepgraphpose_amd import engine, _lib
lib = _lib.load()
N, H, W, Cin, Cout = 1, 8, 16, 32, 128
M = N * H * W
w = np.zeros((1, 1, Cin, Cout), np.float32)
for n in range(Cout):
    w[0, 0, n % Cin, n] = 1.0
wp = torch.from_numpy(engine.pack_conv_weights(w)).cuda()
d = _lib.DgpConvDesc(N, H, W, Cin, Cout, 1, 1, 1, 1, 0, 0, H, W, 0, 0, 0, 0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (m0, k0) in [(0, 0), (0, 16), (8, 0), (4, 8), (1, 3), (20, 25), (40, 9)]:
    x = np.zeros((M, Cin), np.float32); x[m0, k0] = 1.0
    xt = torch.from_numpy(x.reshape(N, H, W, Cin)).cuda()
    rng = torch.zeros(3 * 256, device="cuda")
    _lib.check(lib.dgp_tensor_absmax(C.c_void_p(xt.data_ptr()), xt.numel(), C.c_void_p(rng.data_ptr()), st))
    _lib.check(lib.dgp_tensor_absmax(C.c_void_p(wp.data_ptr()), wp.numel(), C.c_void_p(rng.data_ptr() + 1024), st))
    y = torch.empty((N, H, W, Cout), device="cuda")
    _lib.check(lib.dgp_conv2d_ranged(C.byref(d), C.c_void_p(xt.data_ptr()), C.c_void_p(wp.data_ptr()), None, None, None,
                                     C.c_void_p(y.data_ptr()), C.c_void_p(rng.data_ptr()), C.c_void_p(rng.data_ptr() + 1024),
                                     C.c_void_p(rng.data_ptr() + 2048), st))
    torch.cuda.synchronize()
    g = y.cpu().numpy().reshape(M, Cout)
    nz = np.argwhere(g != 0)
    print("x[%d][%d]=1 -> nonzero outputs (row, col, val):" % (m0, k0), [(int(a), int(b), float(g[a, b])) for a, b in nz][:16])
