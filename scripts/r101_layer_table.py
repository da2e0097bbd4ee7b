"""Per-launch table of one instrumented ResNet-101 1280x720 pass (20 keypoints, batch 16; BASELINE configs[4]'s per-GPU shape)."""
import os, sys, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from deepgraphpose_amd import engine, synthetic
H, W, NJ, B = 720, 1280, 20, 16
net = engine.DGPNet(101, NJ, H, W, max_batch=B); net.load_weights(synthetic.make_weights(101, NJ, False, seed=0))
f = torch.from_numpy(synthetic.make_frames(B, H, W, NJ, seed=1)).cuda()
out = torch.zeros((B, NJ, 5), device="cuda")
for _ in range(3): net.infer_packed(f, out, 1.0, 1)
torch.cuda.synchronize()
net.profile_begin(5)
for _ in range(5): net.infer_packed(f, out, 1.0, 1)
torch.cuda.synchronize()
n, launches = net.profile_end()
agg = collections.OrderedDict()
for name, fl, ms in launches:
    parts = name.replace("conv:resnet_v1_101/", "").replace("bottleneck_v1/", "").split("/")
    key = (parts[0] if parts[0].startswith("block") else name[:30]) + ":" + (parts[-1].split("|")[0] if len(parts) > 1 else "") + "|" + (name.split("|")[1] if "|" in name else "")
    a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += fl; a[2] += ms
tot = sum(v[2] for v in agg.values())
for k, v in agg.items():
    print("%-60s n %3d  %8.3f ms  %6.1f TF  %5.1f %%" % (k[:60], v[0], v[2], v[1] / max(v[2], 1e-9) / 1e9, 100 * v[2] / tot))
print("total", tot)
