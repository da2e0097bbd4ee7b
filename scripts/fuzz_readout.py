"""Random scoremaps through the HIP read-out (soft-argmax + likelihood window, the th branch, the DLC hard arg-max) against the CPU oracle:
map shapes from 1 x 1 to beyond the LDS limit (the streaming path), 1-24 joints, gamma 0.05-20, gauss_len 1-4, peaks on borders and corners,
flat maps, near-ties.  Usage: python scripts/fuzz_readout.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepgraphpose_amd import engine
from oracle import dgp_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = 0
for it in range(n):
    kind = rng.integers(0, 10)
    if kind == 0:
        H, W = int(rng.integers(1, 6)), int(rng.integers(1, 6))                       # tiny maps
    elif kind == 1:
        H, W = int(rng.integers(180, 260)), int(rng.integers(180, 260))               # beyond 38 400 cells: streams from global memory
    else:
        H, W = int(rng.integers(2, 130)), int(rng.integers(2, 170))
    B, C = int(rng.integers(1, 4)), int(rng.integers(1, 25))
    gamma = float(np.exp(rng.uniform(np.log(0.05), np.log(20.0))))
    gl = int(rng.integers(1, 5))
    s = (rng.standard_normal((B, H, W, C)) * rng.uniform(0.1, 3.0)).astype(np.float32)
    mode = rng.integers(0, 5)
    for b in range(B):
        for c in range(C):
            if mode == 0:
                continue                                                              # noise only (broad softmax)
            if mode == 1:
                s[b, :, :, c] = 0.0                                                   # flat map: every window value tied
                continue
            r = int(rng.choice([0, H - 1, rng.integers(0, H)])); q = int(rng.choice([0, W - 1, rng.integers(0, W)]))   # borders / corners / inside
            s[b, r, q, c] += rng.uniform(4.0, 30.0)
            if mode == 3 and H > 1:
                s[b, (r + 1) % H, q, c] = s[b, r, q, c]                               # an exact tie next to the peak
    t = torch.from_numpy(s).cuda()
    mu, conf, idx, pmap = [x.cpu().numpy() for x in engine.soft_argmax(t, gamma, gl, want_pmap=True)]
    mu_ref, pm_ref = O.argmax_2d_from_cm(s, gamma, gl)
    mu64, _ = O.argmax_2d_from_cm(s, gamma, gl, dtype=np.float64)
    e32, e64 = float(np.abs(mu - mu_ref).max() * 8.0), float(np.abs(mu - mu64).max() * 8.0)
    # the fp32 oracle's own distance from float64 grows with the map (sums of up to 60 000 terms): gate on float64, with that slack
    slack = max(1e-3, 1.5 * float(np.abs(mu_ref - mu64).max() * 8.0))
    ep = float(np.abs(pmap - pm_ref).max())
    ok = e64 < slack and ep < 2e-6 and np.isfinite(mu).all()
    for b in range(B):
        iref, lref = O.likelihood_window(s[b], mu[b])
        ok = ok and np.array_equal(idx[b], iref) and float(np.abs(conf[b] - lref).max()) < 2e-6
    th = float(rng.uniform(0.0, 0.9))
    mu_t = engine.pmap_threshold(torch.from_numpy(pmap.copy()).cuda(), th).cpu().numpy()
    mu_t64, _ = O.argmax_2d_from_cm(s, gamma, gl, dtype=np.float64, th=th)
    mu_t32, _ = O.argmax_2d_from_cm(s, gamma, gl, th=th)
    # thresholding is discontinuous: a cell within rounding of th x max may fall on either side.  Judge only the (frame, joint) maps whose
    # closest cell keeps a relative margin of 1e-5 from the threshold in float64 (the kernel's pmap is within ~1e-7 relative of it)
    _, pm64 = O.argmax_2d_from_cm(s, gamma, gl, dtype=np.float64)
    thr = pm64.max(axis=(1, 2), keepdims=True) * th
    margin = (np.abs(pm64 - thr) / np.maximum(thr, 1e-300)).min(axis=(1, 2))              # [B, C]
    clear = margin > 1e-5
    d64 = np.abs(mu_t - mu_t64).max(axis=2) * 8.0                                         # [B, C] px
    et = float(np.where(clear, d64, 0.0).max())
    nborder = int((~clear).sum())
    ok = ok and et < max(1e-3, slack)
    hi, hp, _ = engine.hard_argmax(t)
    for b in range(B):
        sig = O.sigmoid_f32(s[b])
        _, loc_ref = O.argmax_pose_predict(sig, None, 8.0)
        # (sigmoid saturates: several cells may share the maximal fp32 probability; the oracle takes the first of THOSE, the kernel the first
        #  maximum of the logits -- equal unless the map saturates)
        same = np.array_equal(hi[b].cpu().numpy(), loc_ref) or float(s[b].max()) > 15.0
        ok = ok and same
    fails += 0 if ok else 1
    print("%s  %3d x %3d  C %2d B %d gamma %6.2f gl %d mode %d  mu vs fp64 %.1e px (fp32 oracle %.1e)  pmap %.1e  th %.2f: %.1e px (%d borderline maps skipped)"
          % ("ok  " if ok else "FAIL", H, W, C, B, gamma, gl, mode, e64, e32, ep, th, et, nborder), flush=True)
print("failures: %d" % fails)
sys.exit(1 if fails else 0)
