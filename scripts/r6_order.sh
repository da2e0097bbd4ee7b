#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/order
python - "$1" <<'PY' 2>&1 | tail -2
import os, subprocess, sys, numpy as np
code = ("import sys, numpy as np, torch\n"
        "from deepgraphpose_amd import engine\n"
        "from deepgraphpose_amd.synthetic import make_frames, make_weights\n"
        "out = {}\n"
        "for tier in ('parity', 'f16'):\n"
        "    for hw in ((256, 320), (480, 640)):\n"
        "        net = engine.DGPNet(50, 4, hw[0], hw[1], max_batch=8, tier=tier)\n"
        "        net.load_weights(make_weights(50, 4, False, seed=5, head_std=0.05))\n"
        "        mu, conf, idx = net.infer(torch.from_numpy(make_frames(8, hw[0], hw[1], 4, seed=6)).cuda(), 1.0, 1)\n"
        "        out[tier + str(hw[0])] = mu.cpu().numpy()\n"
        "np.savez(sys.argv[1], **out)\n")
res = {}
for lib in ("deepgraphpose_amd/libdgp_hip.so", "build_diag/libdgp_%s.so" % sys.argv[1]):
    path = "/tmp/ord_%s.npz" % os.path.basename(lib)
    subprocess.check_call([sys.executable, "-c", code, path], env=dict(os.environ, DGP_HIP_LIB=lib, PYTHONPATH="."))
    res[lib] = np.load(path)
a, b = res.values()
print("bit-identical with the new tile order:", all(np.array_equal(a[k], b[k]) for k in a.files), list(a.files))
PY
for t in f16 parity; do
  echo "== tier $t"
  timeout 900 bash scripts/ab_tier.sh deepgraphpose_amd/libdgp_hip.so build_diag/libdgp_$1.so $t "block[234]" 2 2>&1 | tail -42
done 2>&1 | tee gpurun_out/order/ab_$1.txt
