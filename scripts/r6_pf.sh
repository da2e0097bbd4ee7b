#!/bin/bash
# A/B of the compute-wave L2 prefetch (-DDGP_PF=<steps>): correctness (bit-identical network outputs), then per-layer times per tier
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/pf
python - <<'PY' 2>&1 | tail -3
import os, subprocess, sys, numpy as np
code = ("import sys, numpy as np, torch\n"
        "from deepgraphpose_amd import engine\n"
        "from deepgraphpose_amd.synthetic import make_frames, make_weights\n"
        "out = {}\n"
        "for tier in ('parity', 'f16'):\n"
        "    net = engine.DGPNet(50, 4, 256, 320, max_batch=4, tier=tier)\n"
        "    net.load_weights(make_weights(50, 4, False, seed=5, head_std=0.05))\n"
        "    mu, conf, idx = net.infer(torch.from_numpy(make_frames(4, 256, 320, 4, seed=6)).cuda(), 1.0, 1)\n"
        "    out[tier] = mu.cpu().numpy()\n"
        "np.savez(sys.argv[1], **out)\n")
res = {}
for lib in ("deepgraphpose_amd/libdgp_hip.so", "build_diag/libdgp_pf3.so"):
    path = "/tmp/pf_%s.npz" % os.path.basename(lib)
    subprocess.check_call([sys.executable, "-c", code, path], env=dict(os.environ, DGP_HIP_LIB=lib, PYTHONPATH="."))
    res[lib] = np.load(path)
a, b = res.values()
print("bit-identical with the prefetch:", all(np.array_equal(a[k], b[k]) for k in ("parity", "f16")))
PY
for t in f16 parity; do
  for v in "$@"; do
    echo "== tier $t, libdgp_$v"
    timeout 900 bash scripts/ab_tier.sh deepgraphpose_amd/libdgp_hip.so build_diag/libdgp_$v.so $t "block[34]/unit_[0-9]/conv1|conv3\+shortcut|block4/unit_./conv3|sum" 2 2>&1 | tail -16
  done
done 2>&1 | tee gpurun_out/pf/ab.txt
