import numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
from deepgraphpose_amd import engine
from deepgraphpose_amd.synthetic import make_frames, make_weights
G = np.load("/root/repo/tests/golden/fullsize_vectors.npz")
wts = make_weights(101, 20, True, seed=41, head_std=0.05)
frames = torch.from_numpy(make_frames(16, 720, 1280, 20, seed=42)).cuda()
net = engine.DGPNet(101, 20, 720, 1280, max_batch=16, with_locref=True)
net.load_weights(wts)
mu, conf, idx = net.infer(frames, 1.0, 1)
d = np.abs(mu.cpu().numpy() - G["r101_mu"]) * 8
print("per-frame max px err", np.round(d.max((1, 2)) * 1e3, 3))
f, j = np.unravel_index(d.max(2).argmax(), d.shape[:2])
print("worst frame", f, "joint", j, d[f, j], "mu", mu[f, j].cpu().numpy(), G["r101_mu"][f, j])
np.save("/root/repo/gpurun_out/r101_mu_gpu.npy", mu.cpu().numpy())
sc = net.forward(frames)
s = sc[f, :, :, j].cpu().numpy()
print("scoremap stats of worst: max", s.max(), "std", s.std(), "softmax peak prob", np.exp(s - s.max()).max() / np.exp(s - s.max()).sum())
