import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np, torch
import torch.nn.functional as F
from test_train_gpu import _train_case
from deepgraphpose_amd.train import Trainer
from deepgraphpose_amd.loss import DGPHyper
from deepgraphpose_amd.arch import resnet_units
from oracle import dgp_train_oracle as T, dgp_oracle as O
batch, S0, wts, frames, ws, ws_max = _train_case(3)
hy = DGPHyper(gm2=1, gm3=3)
# oracle with retained unit outputs
dtype = torch.float64
P = T.make_params(wts, dtype)
name = "resnet_v1_50"
x = torch.from_numpy(frames.astype(np.float32) - np.asarray(O.MEAN_PIXEL, np.float32)).to(dtype).permute(0, 3, 1, 2)
net = F.relu(T._bn(T._conv_same(x, P[name + "/conv1/weights"], 2), P, name + "/conv1"))
_, pt, pb = O.tf_same_pads(net.shape[2], 3, 2); _, pl, pr = O.tf_same_pads(net.shape[3], 3, 2)
net = F.max_pool2d(F.pad(net, (pl, pr, pt, pb), value=float("-inf")), 3, 2)
outs = []
for u in resnet_units(50):
    if u.has_shortcut_conv: sc = T._bn(T._conv(net, P[u.scope + "/shortcut/weights"], u.stride), P, u.scope + "/shortcut")
    else: sc = net if u.stride == 1 else net[:, :, ::u.stride, ::u.stride]
    r = F.relu(T._bn(T._conv(net, P[u.scope + "/conv1/weights"], 1), P, u.scope + "/conv1"))
    r = F.relu(T._bn(T._conv_same(r, P[u.scope + "/conv2/weights"], u.stride, u.rate), P, u.scope + "/conv2"))
    r = T._bn(T._conv(r, P[u.scope + "/conv3/weights"], 1), P, u.scope + "/conv3")
    net = F.relu(sc + r); net.retain_grad(); outs.append(net)
pred = T._deconv(net, P["pose/part_pred/block4/weights"], P["pose/part_pred/block4/biases"]).permute(0, 2, 3, 1)
loc = T._deconv(net, P["pose/locref_pred/block4/weights"], P["pose/locref_pred/block4/biases"]).permute(0, 2, 3, 1)
cfg = dict(nj=3, S0=S0, ws=ws, ws_max=ws_max, stride=8.0, gamma=1.0, gauss_len=1, lengthscale=1.0, gm2=1, gm3=3, wn_visible=5.0,
           wn_hidden=3.0, locref_loss_weight=0.05, locref_huber_loss=True, n_frames_total=300.0, n_visible_frames_total=25.0)
L = T.dgp_loss(pred, loc, batch, cfg); L["total_loss"].backward()
tr = Trainer(50, 3, 64, 96, max_frames=3); tr.load_weights(wts)
for stop in (3,):
    os.environ["DGP_BWD_STOP"] = str(stop); os.environ["DGP_BWD_DUMP"] = "/tmp/g.bin"
    tr.forward_backward(torch.from_numpy(frames).cuda(), batch, hy, S0, ws, ws_max, 300.0, 25.0)
    g = np.fromfile("/tmp/g.bin", dtype=np.float32)
    ui = 15 - stop
    ref = (outs[ui].grad * (outs[ui] > 0)).permute(0, 2, 3, 1).numpy()
    g = g.reshape(ref.shape)
    d = np.abs(g - ref)
    print("stop", stop, "unit", ui, resnet_units(50)[ui].scope[13:], "rel", d.max() / np.abs(ref).max(), "argmax", np.unravel_index(d.argmax(), d.shape), "n bad", (d > 1e-5 * np.abs(ref).max()).sum(), "of", d.size)
    xh = np.fromfile("/tmp/g.bin.x", dtype=np.float32).reshape(ref.shape)
    xo = outs[ui].detach().permute(0, 2, 3, 1).numpy()
    i = np.unravel_index(d.argmax(), d.shape)
    print(" activation at worst element: hip", xh[i], "oracle64", xo[i], "| grad hip", g[i], "ref", ref[i], "| raw oracle grad", outs[ui].grad.permute(0,2,3,1).numpy()[i])
    print(" forward max abs diff", np.abs(xh - xo).max())
    if stop == 3:
        bad = np.argwhere(d > 1e-5 * np.abs(ref).max())
        print(" bad pixels (n,h,w):", sorted(set((int(a), int(b), int(c)) for a, b, c, _ in bad))[:30])
