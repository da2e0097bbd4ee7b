"""N one-stream forward steps of one engine tier on the bench workload, for profilers (every step issues the same dispatch sequence);
writes the per-launch name table of the last steps.  python scripts/run_forward.py parity|f16 [steps=4] [table.tsv] [chain=1]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepgraphpose_amd import engine                                    # noqa: E402
from deepgraphpose_amd.synthetic import make_frames, make_weights       # noqa: E402
tier = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4; table = sys.argv[3] if len(sys.argv) > 3 else ""
B, H, W, NJ = 32, 480, 640, 4
wts = make_weights(50, NJ, False, seed=0, head_std=0.05)
frames = torch.from_numpy(make_frames(B, H, W, NJ, seed=100)).cuda()
net = engine.DGPNet(50, NJ, H, W, max_batch=B, tier=tier)
net.load_weights(wts)
out = torch.zeros((B, NJ, 5), device="cuda")
for _ in range(3):
    net.infer_packed(frames, out)
torch.cuda.synchronize()
net.profile_begin(steps)
for _ in range(steps):
    net.infer_packed(frames, out)
torch.cuda.synchronize()
ns, tab = net.profile_end()
if table:
    with open(table, "w") as f:
        f.write("name\tgflop\tus\n" + "\n".join("%s\t%.3f\t%.2f" % (n, fl / 1e9, ms * 1e3) for n, fl, ms in tab) + "\n")
print("ok", ns, len(tab))
