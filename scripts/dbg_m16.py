import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DGP_CONV2D_CELLS"] = "1"; os.environ["DGP_FORCE_TILE"] = "16"
import ctypes as C
import numpy as np, torch
from deepgraphpose_amd import engine, _lib
lib = _lib.load()
N, H, W, Cin, Cout = 1, 8, 16, 32, 128
M = N * H * W
x = np.zeros((N, H, W, Cin), np.float32)
xm = x.reshape(M, Cin)
for m in range(M):
    xm[m] = m * 64 + np.arange(Cin)            # row id * 64 + k
w = np.zeros((1, 1, Cin, Cout), np.float32)
for n in range(Cout):
    w[0, 0, n % Cin, n] = 1.0                  # out[m][n] = x[m][n % 32]
xt = torch.from_numpy(x).cuda()
wp = torch.from_numpy(engine.pack_conv_weights(w)).cuda()
d = _lib.DgpConvDesc(N, H, W, Cin, Cout, 1, 1, 1, 1, 0, 0, H, W, 0, 0, 0, 0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rng = torch.zeros(3 * 256, device="cuda")
_lib.check(lib.dgp_tensor_absmax(C.c_void_p(xt.data_ptr()), xt.numel(), C.c_void_p(rng.data_ptr()), st))
_lib.check(lib.dgp_tensor_absmax(C.c_void_p(wp.data_ptr()), wp.numel(), C.c_void_p(rng.data_ptr() + 1024), st))
y = torch.empty((N, H, W, Cout), device="cuda")
_lib.check(lib.dgp_conv2d_ranged(C.byref(d), C.c_void_p(xt.data_ptr()), C.c_void_p(wp.data_ptr()), None, None, None,
                                 C.c_void_p(y.data_ptr()), C.c_void_p(rng.data_ptr()), C.c_void_p(rng.data_ptr() + 1024),
                                 C.c_void_p(rng.data_ptr() + 2048), st))
torch.cuda.synchronize()
g = y.cpu().numpy().reshape(M, Cout)
ref = xm[:, np.arange(Cout) % Cin]
print("max err", np.abs(g - ref).max())
np.set_printoptions(linewidth=250, suppress=True)
h = g / 2
print("row 0, all cols (row',k'):")
print([(int(v) // 64, int(v) % 64) for v in h[0]])
print("col 0, all rows (row',k'):")
print([(int(v) // 64, int(v) % 64) for v in h[:, 0]])
print("col 17, rows 0..40:")
print([(int(v) // 64, int(v) % 64) for v in h[:40, 17]])
